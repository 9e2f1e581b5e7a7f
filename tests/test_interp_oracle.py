"""Interpolated-phase plans (ratios without a small rational form) — oracle side, no GPU.

The reference exercises such ratios in tests/test_random.py:21-26 (random integer pairs and random
float pairs in 8000..96000).  Its known-answer test for them is the analytic tone
(/root/reference/tests/test_random.py:139-157, atol 2e-4 because "some rate combination makes error
bigger"; 5 LSB for integers, :160-179); re-expressed here against the oracle with seeded rates.
"""
import random

import numpy as np
import pytest


def tone(freq, rate, seconds):
    n = int(rate * seconds)
    return np.sin(2 * np.pi * freq / rate * np.arange(n)) * np.hanning(n)


def seeded_rate_pairs(seed, n_int=2, n_float=2):
    r = random.Random(seed)
    return ([(r.randint(8000, 96000), r.randint(8000, 96000)) for _ in range(n_int)] +
            [(r.uniform(8000, 96000), r.uniform(8000, 96000)) for _ in range(n_float)])


PAIRS = seeded_rate_pairs(2024)


@pytest.mark.parametrize("in_rate,out_rate", PAIRS)
@pytest.mark.parametrize("quality", ["VHQ", "HQ", "MQ", "LQ", "QQ"])
def test_known_answer_tone_random_rates(oracle, in_rate, out_rate, quality):
    x, want = tone(32.0, in_rate, 4.0), tone(32.0, out_rate, 4.0)
    ref = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
    n = min(len(ref), len(want))
    assert abs(len(ref) - len(want)) <= 1
    assert np.allclose(want[:n], ref[:n], atol=2e-4)
    got = oracle.resample(x.astype(np.float32), in_rate, out_rate, quality, mode="port")
    assert got.dtype == np.float32 and len(got) == len(ref)
    assert np.allclose(want[:n], got[:n], atol=2e-4)
    # canonical f32 order stays within 1e-6 relative RMS of the float64 evaluation
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-6 * np.sqrt(np.mean(ref ** 2))


@pytest.mark.parametrize("in_rate,out_rate", PAIRS)
@pytest.mark.parametrize("dtype", [np.int32, np.int16])
def test_known_answer_tone_int_random_rates(oracle, in_rate, out_rate, dtype):
    x = (tone(32.0, in_rate, 4.0) * 16384).astype(dtype)
    want = (tone(32.0, out_rate, 4.0) * 16384).astype(dtype)
    got = oracle.resample(x, in_rate, out_rate, "HQ", mode="port")
    n = min(len(got), len(want))
    assert got.dtype == dtype
    assert np.allclose(want[:n], got[:n], atol=5)   # the reference's tolerance for random rates (:177)


def test_interp_and_exact_plans_agree_where_both_exist(oracle):
    """A ratio just under the exact-bank limit and its interpolated evaluation describe the same
    filter: evaluate the cubic table of a neighbouring interpolated plan at the exact plan's phases."""
    ex = oracle.plan(48000, 44100, "VHQ")            # exact: L = 147
    ip = oracle.plan(48000, 44101, "VHQ")            # interpolated, same band edges to 2e-5
    assert ex.phases == 0 and ip.phases == 128 and ex.T == ip.T
    for ph in (0, 1, 73, 146):
        f = ph / ex.L
        c = ip.exact_coefs(f)
        # same prototype up to the (tiny) change of cut-off between 44100 and 44101
        assert np.abs(c - ex.bank[ph]).max() < 2e-5


def test_interp_position_is_chunk_invariant(oracle):
    pl = oracle.plan(44100.123456789, 47999.987654321, "HQ")
    assert pl.phases == 32
    rng = np.random.default_rng(3)
    x = rng.standard_normal(5000).astype(np.float32)
    whole = oracle.resample_channel(pl, x, "port_f32")
    # second half of the outputs from a window of the input that starts at absolute index 1000
    k0 = len(whole) // 2
    part = oracle.resample_channel(pl, x[1000:], "port_f32", k0=k0, n_out=len(whole) - k0, in_abs0=1000)
    lo = int(np.ceil((1000 + pl.T) * pl.L / pl.M)) + 1   # outputs whose taps lie inside the window
    assert lo < len(whole)
    assert np.array_equal(part[max(lo - k0, 0):], whole[max(lo, k0):])
