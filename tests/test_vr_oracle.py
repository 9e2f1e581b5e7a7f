"""Variable rate — oracle side (no GPU).  The reference does not test this mode (SURVEY.md §8(f)-3;
its only use is the manual script tests/vr.py), so what can be asserted is analytic: a sine resampled
on a moving clock must equal the sine evaluated at the clock's positions, the clock must be
continuous and strictly increasing across ratio changes, and a constant-ratio VR stream must agree
with the fixed-ratio resampler."""
import numpy as np
import pytest

from vr_sim import ONE, VrSim, q64


def test_constant_ratio_vr_matches_fixed_ratio_resampler(oracle):
    in_rate, out_rate = 48000, 16000
    sim = VrSim(oracle, in_rate, out_rate, "HQ", np.float64)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(20000) * 0.25
    y = np.concatenate([sim.feed(x[:7000], mode="ref"), sim.feed(x[7000:], last=True, mode="ref")])
    # the fixed-ratio plan for 3:1 is an exact bank: same prototype, no coefficient interpolation
    want = oracle.resample(x, in_rate, out_rate, "HQ", mode="ref")
    assert len(y) == len(want)
    assert np.sqrt(np.mean((y - want) ** 2)) <= 2.0 ** -21 * np.sqrt(np.mean(want ** 2))   # HQ: 20 bits


@pytest.mark.parametrize("quality,tol", [("VHQ", 2e-6), ("HQ", 2e-6), ("MQ", 2e-5), ("LQ", 2e-4)])
def test_sine_on_a_slewing_clock_is_the_sine_at_the_clock_positions(oracle, quality, tol):
    fs = 48000.0
    sim = VrSim(oracle, fs, 16000.0, quality, np.float64)     # largest io ratio 3
    n = np.arange(60000)
    f0 = 440.0
    x = np.sin(2 * np.pi * f0 / fs * n)
    got, marks = [], []
    got.append(sim.feed(x[:15000], mode="ref"))
    sim.set_io_ratio(1.5, 2000)                               # slew 3 -> 1.5 over 2000 outputs
    marks.append(sim.k_done)
    got.append(sim.feed(x[15000:30000], mode="ref"))
    sim.set_io_ratio(2.25, 0)                                 # jump
    marks.append(sim.k_done)
    got.append(sim.feed(x[30000:45000], mode="ref"))
    sim.set_io_ratio(3.0, 777)
    got.append(sim.feed(x[45000:], last=True, mode="ref"))
    y = np.concatenate(got)
    # replay the clock
    clock = VrSim(oracle, fs, 16000.0, quality, np.float64)
    t = []
    k = 0
    for upto, change in [(marks[0], (1.5, 2000)), (marks[1], (2.25, 0)), (sim.k_done - len(got[3]), (3.0, 777)),
                         (len(y), None)]:
        t.append(clock.positions(k, upto))
        k = clock.k_done = upto
        if change:
            clock.set_io_ratio(*change)
    t = np.concatenate(t)
    assert len(t) == len(y)
    assert np.all(np.diff(t) > 0)                             # strictly increasing
    steps = np.diff(t)
    assert np.abs(np.diff(steps)).max() <= 0.76               # only the one deliberate jump (1.5 -> 2.25)
    inside = (t > sim.vp.T) & (t < len(x) - sim.vp.T)         # away from the zero-extended ends
    want = np.sin(2 * np.pi * f0 / fs * t)
    assert np.abs(y[inside] - want[inside]).max() <= tol
    # output count: the clock passes all input
    assert 0 <= len(x) - t[-1] <= 1.5 * 3.0                    # within 1.5 steps of the end of the input


def test_q64_and_positions_are_exact_integers():
    assert q64(3.0) == 3 * ONE and q64(0.5) == ONE // 2
    assert q64(1 / 3) == int((1 / 3) * 2 ** 64)               # exact scaling of the double, truncated
