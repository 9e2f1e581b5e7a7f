"""Host logic of the product library (no GPU needed): ratio reduction, filter design, bank layout,
length rule — checked through the C ABI against the oracle and against the recipe's spec."""
import ctypes as C

import numpy as np
import pytest

RATES = [(48000, 44100), (44100, 16000), (44100, 32000), (32000, 44100), (48000, 22050), (8000, 48000),
         (44100, 22050), (22050, 32000), (100, 200), (48000, 24000), (96000, 44100), (100.5, 200)]
QUALS = ["VHQ", "HQ", "MQ", "LQ", "QQ"]


@pytest.mark.parametrize("in_rate,out_rate", RATES)
@pytest.mark.parametrize("quality", QUALS)
def test_bank_matches_independent_design(oracle, in_rate, out_rate, quality):
    """Two independently written implementations of the same specification — plan.cpp (C++, scalar
    loops, power-series I0) and oracle/design.py (numpy/scipy, Chebyshev I0, vectorised layout) —
    agree on the geometry exactly and on the float64 bank to 2e-13 of its scale."""
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality)
    o = oracle.plan(in_rate, out_rate, quality)
    assert (p.L, p.M, p.taps) == (o.L, o.M, o.T)
    assert p.taps % 8 == 0
    assert np.abs(p.bank() - o.bank).max() <= 2e-13 * np.abs(o.bank).max()


def _response(plan, nfft=1 << 22):
    L, T = plan.L, plan.taps
    bank = plan.bank()
    g = np.zeros(L * T)
    for ph in range(L):
        g[L * (T - 1 - np.arange(T)) + ph] = bank[ph]
    H = np.abs(np.fft.rfft(g, nfft)) / L
    f = np.arange(H.size) * (L * plan.in_rate) / nfft
    return f, H


@pytest.mark.parametrize("in_rate,out_rate", [(48000, 44100), (44100, 16000), (32000, 44100), (8000, 48000)])
@pytest.mark.parametrize("quality,bits", [("VHQ", 28), ("HQ", 20), ("MQ", 16), ("LQ", 16)])
def test_frequency_response_meets_recipe(in_rate, out_rate, quality, bits):
    """Pass-band flat to the recipe's precision up to passband_end * Nyquist(lower rate); stop-band
    (from Nyquist of the lower rate) down by at least (bits+1)*6.02 dB."""
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality)
    f, H = _response(p)
    fn = min(in_rate, out_rate) / 2
    passband = H[f <= p.passband_end * fn]
    stop = H[f >= p.stopband_begin * fn]
    att_spec = (bits + 1) * 20 * np.log10(2)
    assert 20 * np.log10(stop.max()) <= -att_spec
    ripple = max(passband.max() - 1, 1 - passband.min())
    assert ripple <= 2.0 ** -(bits - 1)
    # each phase has unit DC gain to well below the precision
    assert np.abs(p.bank().sum(axis=1) - 1).max() <= 2.0 ** -(bits + 2)


def test_plan_info_and_lengths():
    from soxr_amd import device as dev
    p = dev.Plan(48000, 44100, "VHQ")
    assert (p.L, p.M) == (147, 160)
    assert abs(p.passband_end - 0.91151) < 1e-4 and p.stopband_begin == 1.0 and p.precision_bits == 28
    for n, want in [(0, 0), (1, 1), (480000, 441000), (2880000, 2646000), (159, 146), (160, 147), (161, 148)]:
        assert p.out_len(n) == want == int(np.floor(n * 147 / 160 + 0.5))
    p2 = dev.Plan(100.5, 200, "HQ")
    assert (p2.L, p2.M) == (400, 201)


def test_set_bank_round_trip():
    from soxr_amd import device as dev
    p = dev.Plan(44100, 16000, "HQ")
    b = p.bank()
    b2 = b * 0.5
    p.set_bank(b2)
    assert np.array_equal(p.bank(), b2)
    with pytest.raises(RuntimeError):
        p.set_bank(np.zeros(7))


def test_invalid_plans():
    from soxr_amd import device as dev
    with pytest.raises(ValueError):
        dev.Plan(0, 44100)
    with pytest.raises(ValueError):
        dev.Plan(48000, 44100, "best")


# ratios without a small rational form (reference tests/test_random.py:21-26: random integer and
# float rates) -> interpolated-phase plans
INTERP_RATES = [(48000, 44101), (44100.123456789, 47999.987654321), (95999, 8001), (8000.5, 96000.25),
                (12345.678, 54321.9), (77777, 33333.3),
                # a ratio whose continued fraction leaves the 31-bit range before reaching 1e-15: the
                # best semiconvergent inside the range stands in (found by tests/fuzz/fuzz_vs_oracle.py)
                (51387.21808175107, 37891.91119494756)]


@pytest.mark.parametrize("in_rate,out_rate", INTERP_RATES)
@pytest.mark.parametrize("quality", QUALS)
def test_interp_plan_matches_independent_design(oracle, in_rate, out_rate, quality):
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality)
    o = oracle.plan(in_rate, out_rate, quality)
    assert (p.L, p.M, p.taps, p.phases) == (o.L, o.M, o.T, o.phases)
    assert abs(p.L / p.M - out_rate / in_rate) <= 4e-15 * out_rate / in_rate and 0 < p.L < 2 ** 31 and 0 < p.M < 2 ** 31
    assert np.abs(p.bank() - o.bank).max() <= 1e-12 * np.abs(o.bank).max()
    if p.phases:
        assert p.L * p.taps > 1 << 22
        n = 123457
        assert p.out_len(n) == o.out_len(n)
        assert abs(p.out_len(n) - n * out_rate / in_rate) <= 0.5 + 1e-6


@pytest.mark.parametrize("in_rate,out_rate", [(48000, 44101), (95999, 8001), (12345.678, 54321.9)])
@pytest.mark.parametrize("quality,bits", [("VHQ", 28), ("HQ", 20), ("MQ", 16)])
def test_interp_table_accuracy(oracle, in_rate, out_rate, quality, bits):
    """The cubic table reproduces the continuous prototype to far below the recipe's precision, and
    every fractional position has unit DC gain."""
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality)
    assert p.phases > 0
    tab = p.bank()
    o = oracle.plan(in_rate, out_rate, quality)
    rng = np.random.default_rng(5)
    worst = dc = 0.0
    for f in np.concatenate([rng.random(24), [0.0, 1 - 2.0 ** -30]]):
        iv = int(f * p.phases)
        x = f * p.phases - iv
        a = tab[iv]
        c = a[:, 0] + x * (a[:, 1] + x * (a[:, 2] + x * a[:, 3]))
        worst = max(worst, np.abs(c - o.exact_coefs(f)).max())
        dc = max(dc, abs(c.sum() - 1))
    assert worst <= 2.0 ** -(bits + 4)
    assert dc <= 2.0 ** -(bits + 1)
