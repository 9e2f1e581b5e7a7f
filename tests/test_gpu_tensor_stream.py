"""GPU: soxr_amd.device.TensorStream — the device-resident counterpart of ResampleStream (reference surface
src/soxr/__init__.py:56-131; the stream invariances of tests/test_resample.py:60-110: any cut of the signal into calls
gives the one-shot result).  Chunks are device tensors, the state stays in HBM, a call is one asynchronous launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sig(rng, shape, dtype):
    if np.issubdtype(dtype, np.integer):
        return (rng.standard_normal(shape) * 5000).astype(dtype)
    return (rng.standard_normal(shape) * 0.25).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
@pytest.mark.parametrize("rates,quality,ch", [((44100, 16000), "VHQ", 1), ((48000, 44100), "HQ", 2), ((16000, 48000), "MQ", 3),
                                              ((48000, 44101.5), "HQ", 2)])
def test_tensor_stream_equals_oneshot_and_host_stream(soxr, dtype, rates, quality, ch):
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(21)
    x = _sig(rng, (40000, ch) if ch > 1 else (40000,), dtype)
    want = soxr.resample(x, rates[0], rates[1], quality=quality)                 # host one-shot (oracle-checked elsewhere)
    xd = torch.from_numpy(x).cuda()
    ts = dev.TensorStream(rates[0], rates[1], ch, dtype=xd.dtype, quality=quality)
    rs = soxr.ResampleStream(rates[0], rates[1], ch, dtype=dtype, quality=quality)
    cuts = [0, 17, 17, 500, 4410, 4411, 12000, 12000, 30000, 39999, 40000]       # an empty call, tiny, mid, large
    got, host = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        last = b == 40000
        y = ts.resample_chunk(xd[a:b], last=last)
        assert y.is_cuda and y.dtype == xd.dtype and (y.ndim == xd.ndim)
        got.append(y.cpu().numpy())
        host.append(rs.resample_chunk(x[a:b], last=last))
        if dtype != np.int16:  # (int16: the host stream dithers by default, the device job only on request)
            assert np.array_equal(got[-1], host[-1]), (a, b)                     # same frames in the same calls
    got = np.concatenate(got)
    if dtype == np.int16:
        xd_all = dev.resample_tensor(dev.Plan(rates[0], rates[1], quality), xd, kernel=dev.KERNEL_EXACT).cpu().numpy()
        assert np.array_equal(got, xd_all)
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 2    # dither: within an LSB or two
    else:
        assert got.shape == want.shape and np.array_equal(got, want)
    assert ts.delay() < 1.0
    with pytest.raises(RuntimeError):
        ts.resample_chunk(xd[:10])                                               # input after the last input
    ts.clear()
    again = ts.resample_chunk(xd, last=True).cpu().numpy()
    assert np.array_equal(again, got)


def test_tensor_stream_long_run_retires_input(soxr):
    """Thousands of small calls: the device ring is compacted, never grows without bound, and the result is the
    one-shot result."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(22)
    x = _sig(rng, (441 * 700,), np.float32)
    xd = torch.from_numpy(x).cuda()
    ts = dev.TensorStream(44100, 16000, 1, dtype=torch.float32, quality="VHQ")
    parts = [ts.resample_chunk(xd[a:a + 441], last=(a + 441 >= len(x))) for a in range(0, len(x), 441)]
    got = torch.cat(parts).cpu().numpy()
    assert np.array_equal(got, soxr.resample(x, 44100, 16000, quality="VHQ"))
    assert ts._buf.shape[0] <= 8192


def test_tensor_stream_clip_counter_and_errors(soxr):
    import torch
    from soxr_amd import device as dev
    x = torch.full((6000,), 32767, dtype=torch.int16, device="cuda")
    x[(torch.arange(6000, device="cuda") // 100) % 2 == 1] = -32768               # full-scale square wave: the filter overshoots
    ts = dev.TensorStream(48000, 44100, 1, dtype=torch.int16, quality="VHQ")
    ts.resample_chunk(x, last=True)
    rs = soxr.ResampleStream(48000, 44100, 1, dtype="int16", quality="VHQ")
    rs.resample_chunk(x.cpu().numpy(), last=True)
    assert ts.num_clips() > 0 and abs(ts.num_clips() - rs.num_clips()) <= rs.num_clips() // 10 + 2   # (the host stream dithers)
    with pytest.raises(TypeError):
        dev.TensorStream(48000, 44100, 1, dtype=torch.int16).resample_chunk(x.float())
    with pytest.raises(RuntimeError):
        dev.TensorStream(48000, 44100, 1, dtype=torch.float32).resample_chunk(torch.zeros(10))
    with pytest.raises(ValueError):
        dev.TensorStream(48000, 44100, 2, dtype=torch.float32).resample_chunk(torch.zeros(10, device="cuda"))
