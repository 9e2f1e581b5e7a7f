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
        assert np.array_equal(got[-1], host[-1]), (a, b)                         # same frames in the same calls
    got = np.concatenate(got)
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


def test_tensor_stream_clip_counter_and_errors(soxr):
    import torch
    from soxr_amd import device as dev
    x = torch.full((6000,), 32767, dtype=torch.int16, device="cuda")
    x[(torch.arange(6000, device="cuda") // 100) % 2 == 1] = -32768               # full-scale square wave: the filter overshoots
    ts = dev.TensorStream(48000, 44100, 1, dtype=torch.int16, quality="VHQ")
    ts.resample_chunk(x, last=True)
    rs = soxr.ResampleStream(48000, 44100, 1, dtype="int16", quality="VHQ")
    rs.resample_chunk(x.cpu().numpy(), last=True)
    assert ts.num_clips() > 0 and ts.num_clips() == rs.num_clips()
    with pytest.raises(TypeError):
        dev.TensorStream(48000, 44100, 1, dtype=torch.int16).resample_chunk(x.float())
    with pytest.raises(RuntimeError):
        dev.TensorStream(48000, 44100, 1, dtype=torch.float32).resample_chunk(torch.zeros(10))
    with pytest.raises(ValueError):
        dev.TensorStream(48000, 44100, 2, dtype=torch.float32).resample_chunk(torch.zeros(10, device="cuda"))


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.float64])
def test_tensor_stream_variable_rate_equals_host_stream(soxr, dtype):
    """vr=True: the stream handle's own Q64.64 clock serves device chunks too — the same frames in the same calls as the host
    stream (which tests/test_gpu_vr.py pins to the oracle), across jumps, slews, a change during a slew, small and large
    chunks (k_chain / k_interp_wave)."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(23)
    x = _sig(rng, (160000, 2), dtype)
    xd = torch.from_numpy(x).cuda()
    ts = dev.TensorStream(44100, 16000, 2, dtype=xd.dtype, quality="VHQ", vr=True)
    rs = soxr.ResampleStream(44100, 16000, 2, dtype=dtype, quality="VHQ", vr=True)
    cuts = [0, 441, 882, 5000, 45000, 45000, 46000, 90000, 120000, 160000]
    changes = {2: (44100, 22050, 300), 3: (5, 2, 0), 5: (44100, 30000, 4000), 6: (44100, 16000, 50), 7: (1, 1, 1000)}
    total = 0
    for c, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if c in changes:
            ts.set_io_ratio(*changes[c])
            rs.set_io_ratio(*changes[c])
        last = b == cuts[-1]
        y = ts.resample_chunk(xd[a:b], last=last).cpu().numpy()
        w = rs.resample_chunk(x[a:b], last=last)
        assert y.shape == w.shape and np.array_equal(y, w), (c, a, b)
        total += len(y)
    assert total > 60000 and ts.delay() < 2
    with pytest.raises(RuntimeError):
        dev.TensorStream(44100, 16000, 1, dtype=torch.float32).set_io_ratio(2, 1)   # needs vr=True


def test_tensor_stream_on_a_side_stream_and_mixed_with_host_calls(soxr):
    """A call is ordered on the CURRENT torch stream; host-pointer calls on the same handle (its own HIP stream) are
    ordered against device calls by events."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(24)
    x = _sig(rng, (60000,), np.float32)
    want = soxr.resample(x, 48000, 44100, quality="VHQ")
    side = torch.cuda.Stream()
    ts = dev.TensorStream(48000, 44100, 1, dtype=torch.float32, quality="VHQ")
    parts = []
    with torch.cuda.stream(side):
        xd = torch.from_numpy(x).cuda()
        for a in range(0, 60000, 7000):
            parts.append(ts.resample_chunk(xd[a:a + 7000], last=(a + 7000 >= 60000)))
        y = torch.cat(parts)
    side.synchronize()
    assert np.array_equal(y.cpu().numpy(), want)
