"""GPU: k_gather_wave (round 4) — exact-bank launches of 4096 outputs and more that do not go to the period-tile kernels: a
half-chain per quad of lanes on the phase-major bank.  Same canonical arithmetic: bit-identical to the tile kernels and to
the oracle, for every dtype, for half-chains that are and are not multiples of sixteen taps, for one and several columns,
at the ends of the signal (zero extension), and inside streams whose chunks take it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sig(rng, shape, dtype):
    if np.issubdtype(dtype, np.integer):
        return (rng.standard_normal(shape) * 5000).astype(dtype)
    return (rng.standard_normal(shape) * 0.25).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
@pytest.mark.parametrize("rates,quality,ch", [((48000, 44100), "VHQ", 1),      # T = 296: half-chains of 148 = 9 x 16 + 4 taps
                                              ((44100, 16000), "VHQ", 2),      # T = 736: 23 x 16
                                              ((44100, 48000), "HQ", 3), ((16000, 48000), "MQ", 1), ((96000, 44100), "LQ", 2)])
def test_forced_gather_path_equals_tiles_and_oracle(soxr, oracle, dtype, rates, quality, ch):
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(31)
    x = _sig(rng, (12000, ch) if ch > 1 else (12000,), dtype)
    plan = dev.Plan(rates[0], rates[1], quality)
    xd = torch.from_numpy(x).cuda()
    yg = dev.resample_tensor(plan, xd, kernel=dev.KERNEL_GATHER).cpu().numpy()      # >= 4096 outputs: k_gather_wave
    ye = dev.resample_tensor(plan, xd, kernel=dev.KERNEL_EXACT).cpu().numpy()       # AUTO among the exact kernels: tiles
    assert np.array_equal(yg, ye)
    if ch == 1 and dtype != np.int16:   # (int16: the oracle's one-shot dithers, the device job does not)
        assert np.array_equal(yg, oracle.resample(x, rates[0], rates[1], quality))


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.float64, np.int32])
def test_stream_chunks_on_gather_wave_equal_oneshot(soxr, dtype):
    """20 000-frame chunks of a host-array stream (results written straight into pinned host memory) take k_gather_wave; the
    concatenation is the one-shot result (tile kernels; oracle-checked in test_gpu_parity.py)."""
    rng = np.random.default_rng(32)
    x = _sig(rng, (90001, 2), dtype)
    want = soxr.resample(x, 44100, 16000, quality="VHQ")
    rs = soxr.ResampleStream(44100, 16000, 2, dtype=dtype, quality="VHQ")
    got = np.concatenate([rs.resample_chunk(x[a:a + 20000], last=(a + 20000 >= len(x))) for a in range(0, len(x), 20000)])
    assert got.shape == want.shape and np.array_equal(got, want)
