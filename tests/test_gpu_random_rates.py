"""GPU: arbitrary (random integer / random float) rate pairs — the reference's tests/test_random.py,
re-expressed with seeded rates so that runs are reproducible, plus bit-exact parity of the
interpolated-phase kernel (k_interp) with the oracle.

Reference counterparts: get_random_sr_pairs (test_random.py:21-26), test_divide_match (:41-53),
test_length_match (:56-68), test_stream_length (:103-114), test_stream_int (:117-128),
test_quality_sine (:139-157), test_int_sine (:160-178).
"""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def seeded_rate_pairs(seed, n_int=2, n_float=2):
    r = random.Random(seed)
    return ([(r.randint(8000, 96000), r.randint(8000, 96000)) for _ in range(n_int)] +
            [(r.uniform(8000, 96000), r.uniform(8000, 96000)) for _ in range(n_float)])


PAIRS = seeded_rate_pairs(2024)          # the pairs tests/test_interp_oracle.py pins the oracle on
PAIRS_B = seeded_rate_pairs(77)
DTYPES = [np.float32, np.float64, np.int16, np.int32]


def tone(freq, rate, seconds):
    n = int(rate * seconds)
    return np.sin(2 * np.pi * freq / rate * np.arange(n)) * np.hanning(n)


def _signal(rng, n, ch, dtype):
    if np.issubdtype(dtype, np.integer):
        return (rng.standard_normal((n, ch)) * 5000).astype(dtype)
    return (rng.standard_normal((n, ch)) * 0.25).astype(dtype)


def _stream(soxr, x, in_rate, out_rate, chunk, dtype, quality="HQ"):
    rs = soxr.ResampleStream(in_rate, out_rate, x.shape[1], dtype=dtype, quality=quality)
    parts = [np.empty((0, x.shape[1]), dtype)]
    if len(x) == 0:
        parts.append(rs.resample_chunk(x, last=True))
    for i in range(0, len(x), chunk):
        parts.append(rs.resample_chunk(x[i:i + chunk], last=(i + chunk >= len(x))))
    return np.concatenate(parts)


# ---- parity with the oracle (bit-exact, every dtype and recipe) ---------------------------------
@pytest.mark.parametrize("in_rate,out_rate", PAIRS + PAIRS_B[2:])
@pytest.mark.parametrize("dtype", DTYPES)
def test_interp_bit_exact_vs_port(soxr, oracle, in_rate, out_rate, dtype):
    rng = np.random.default_rng(4321)
    x = _signal(rng, 5000, 2, dtype)
    for q in ("VHQ", "HQ", "MQ", "LQ", "QQ"):
        y = soxr.resample(x, in_rate, out_rate, quality=q)
        want = oracle.resample(x, in_rate, out_rate, q, mode="port")
        assert y.dtype == x.dtype and y.shape == want.shape
        assert np.array_equal(y, want), f"{q}: max diff {np.abs(y.astype(np.float64) - want).max()}"
        if np.issubdtype(dtype, np.floating):
            ref = oracle.resample(x, in_rate, out_rate, q, mode="ref")
            err = np.sqrt(np.mean((y - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
            assert err <= 1e-6


def test_interp_device_job_layouts_and_kernels(oracle):
    import torch
    from soxr_amd import device as dev
    in_rate, out_rate = PAIRS[2]
    plan = dev.Plan(in_rate, out_rate, "VHQ")
    assert plan.phases == 128
    rng = np.random.default_rng(8)
    x = _signal(rng, 40000, 3, np.float32)
    want = oracle.resample(x, in_rate, out_rate, "VHQ", mode="port")
    xt = torch.from_numpy(x).cuda()
    for kernel in (dev.KERNEL_GATHER, dev.KERNEL_EXACT):
        assert np.array_equal(dev.resample_tensor(plan, xt, kernel=kernel).cpu().numpy(), want)
    # AUTO / KERNEL_FFT on a float job of this size: the two-stage form (round 4; 1e-6 class, tests/test_gpu_two_stage.py)
    rms = lambda v: float(np.sqrt(np.mean(np.square(v, dtype=np.float64))))
    for kernel in (dev.KERNEL_AUTO, dev.KERNEL_FFT):
        got = dev.resample_tensor(plan, xt, kernel=kernel).cpu().numpy()
        assert got.shape == want.shape and rms(got - want) <= 1e-6 * rms(want)
    for kernel in (dev.KERNEL_TILE, dev.KERNEL_TILE_MFMA):
        with pytest.raises(RuntimeError):
            dev.resample_tensor(plan, xt, kernel=kernel)
    # batch of clips, planar
    xb = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()[:, :, None]       # [clips=3, frames, 1]
    yb = dev.resample_tensor(plan, xb, kernel=dev.KERNEL_EXACT).cpu().numpy()[:, :, 0]
    assert np.array_equal(yb, want.T)
    ya = dev.resample_tensor(plan, xb).cpu().numpy()[:, :, 0]
    assert rms(ya - want.T) <= 1e-6 * rms(want)


# ---- the reference's invariances on random rates -----------------------------------------------
@pytest.mark.parametrize("in_rate,out_rate", PAIRS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_divide_match(soxr, in_rate, out_rate, dtype):
    x = np.random.default_rng(11).standard_normal((31237, 2)).astype(dtype)
    one = soxr._resample_oneshot(x, in_rate, out_rate)
    assert np.array_equal(one, soxr.resample(x, in_rate, out_rate))
    assert np.array_equal(one, soxr.resample(np.asfortranarray(x), in_rate, out_rate))
    assert np.array_equal(one, soxr._resample_divided(x, in_rate, out_rate))


@pytest.mark.parametrize("in_rate,out_rate", PAIRS_B)
@pytest.mark.parametrize("length", [0, 1, 2, 4099, 77001, 149999])
def test_length_match(soxr, in_rate, out_rate, length):
    x = np.random.default_rng(12).standard_normal((163841, 2)).astype(np.float32)
    one = soxr._resample_oneshot(x[:length], in_rate, out_rate)
    assert abs(len(one) - length * out_rate / in_rate) <= 0.5 + 1e-6
    assert np.array_equal(one, soxr.resample(x[:length], in_rate, out_rate))
    assert np.array_equal(one, soxr.resample(np.asfortranarray(x)[:length], in_rate, out_rate))


@pytest.mark.parametrize("in_rate,out_rate", PAIRS)
@pytest.mark.parametrize("chunk", [23, 20011])
@pytest.mark.parametrize("length", [0, 3, 50021, 120007])
@pytest.mark.parametrize("dtype", ["float32", np.float64])
def test_stream_length(soxr, in_rate, out_rate, chunk, length, dtype):
    if chunk == 23 and length > 60000:
        length = 8009  # tiny chunks: keep the number of host round trips bounded
    x = np.random.default_rng(13).standard_normal((length, 1)).astype(dtype)
    assert np.array_equal(soxr._resample_oneshot(x, in_rate, out_rate),
                          _stream(soxr, x, in_rate, out_rate, chunk, np.dtype(dtype)))


@pytest.mark.parametrize("in_rate,out_rate", PAIRS_B)
@pytest.mark.parametrize("chunk", [37, 4999])
@pytest.mark.parametrize("length", [1, 7001, 29989])
@pytest.mark.parametrize("dtype", ["int32", np.int16])
def test_stream_int(soxr, in_rate, out_rate, chunk, length, dtype):
    if chunk == 37:
        length = min(length, 7001)
    x = (np.random.default_rng(14).standard_normal((length, 2)) * 5000).astype(dtype)
    one = soxr._resample_oneshot(x, in_rate, out_rate)
    st = _stream(soxr, x, in_rate, out_rate, chunk, np.dtype(dtype))
    assert np.allclose(one, st, atol=2)      # the reference's bar
    assert np.array_equal(one, st)           # ours


@pytest.mark.parametrize("in_rate,out_rate", PAIRS + PAIRS_B)
@pytest.mark.parametrize("quality", ["VHQ", "HQ", "SOXR_MQ", "lq", "soxr_qq"])
def test_quality_sine(soxr, in_rate, out_rate, quality):
    x, want = tone(32.0, in_rate, 4.0), tone(32.0, out_rate, 4.0)
    got = soxr.resample(x, in_rate, out_rate, quality=quality)
    split = soxr.resample(np.asfortranarray(x), in_rate, out_rate, quality=quality)
    n = min(len(want), len(got))
    assert np.allclose(want[:n], got[:n], atol=2e-4)      # the reference's tolerance for this test
    assert np.allclose(want[:n], split[:n], atol=2e-4)


@pytest.mark.parametrize("in_rate,out_rate", PAIRS + PAIRS_B)
@pytest.mark.parametrize("dtype", [np.int32, np.int16])
def test_int_sine(soxr, in_rate, out_rate, dtype):
    x = (tone(32.0, in_rate, 4.0) * 16384).astype(dtype)
    want = (tone(32.0, out_rate, 4.0) * 16384).astype(dtype)
    got = soxr.resample(x, in_rate, out_rate)
    split = soxr.resample(np.asfortranarray(x), in_rate, out_rate)
    one = soxr._resample_oneshot(x, in_rate, out_rate)
    n = min(len(want), len(got))
    # the reference's tolerances for random rates (test_random.py:176-179)
    assert np.allclose(want[:n], got[:n], atol=5)
    assert np.allclose(want[:n], split[:n], atol=5)
    assert np.allclose(one, split, atol=2)
    assert np.array_equal(one, split)
