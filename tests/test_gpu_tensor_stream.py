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


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
@pytest.mark.parametrize("ch", [1, 2, 3])
def test_tensor_stream_large_chunks_of_an_irrational_ratio(soxr, dtype, ch):
    """Round 5: calls large enough for the interpolated-phase tile kernel, whose lanes carry two outputs (the neighbouring
    channel, or the column's second half a whole number of periods on — here 180 000-frame calls are 3.75 periods of 44101
    outputs).  Stream position, ring start and dither context differ from call to call; the frames must still be the
    one-shot's, bit for bit (reference contract tests/test_resample.py:105-116)."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(5 + ch)
    n = 400000
    x = _sig(rng, (n, ch) if ch > 1 else (n,), dtype)
    want = soxr.resample(x, 48000, 44101, quality="VHQ")
    xd = torch.from_numpy(x).cuda()
    ts = dev.TensorStream(48000, 44101, ch, dtype=xd.dtype, quality="VHQ")
    cuts = [0, 180000, 360000, 360017, n]
    got = [ts.resample_chunk(xd[a:b], last=(b == n)).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:])]
    got = np.concatenate(got)
    assert got.shape == want.shape and np.array_equal(got, want)


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


@pytest.mark.parametrize("dtype,rates,quality,ch", [(np.int16, (44100, 16000), "VHQ", 1), (np.float32, (48000, 44100), "HQ", 2),
                                                    (np.float32, (48000, 44101.5), "HQ", 1), (np.int32, (16000, 48000), "MQ", 2)])
def test_stream_group_one_launch_equals_streams_called_singly(soxr, dtype, rates, quality, ch):
    """`TensorStreamGroup` / hipsoxr_streams_process_device: N INDEPENDENT handles — different phases, different pending
    counts (every stream is fed a prefix of its own length first), distinct dither seeds — served by one launch per call.
    Stream i's frames must be the frames a TensorStream of its own returns for the same chunks, bit for bit, and a group
    member stays an ordinary stream (used alone in between, flushed alone at the end)."""
    import torch
    from soxr_amd import device as dev
    n, chunk, rounds = 37, 441, 12
    rng = np.random.default_rng(31)
    tdt = torch.from_numpy(np.zeros(1, dtype)).dtype
    seeds = list(range(100, 100 + n))
    grp = dev.TensorStreamGroup(n, rates[0], rates[1], ch, dtype=tdt, quality=quality, dither_seeds=seeds)
    solo = [dev.TensorStream(rates[0], rates[1], ch, dtype=tdt, quality=quality, dither_seed=seeds[i]) for i in range(n)]
    shape = lambda f: (f, ch) if ch > 1 else (f,)            # noqa: E731
    got = [[] for _ in range(n)]
    want = [[] for _ in range(n)]
    for i in range(n):                                       # de-phase the streams: prefixes of 0 .. 36 * 53 frames
        pre = torch.from_numpy(_sig(rng, shape(i * 53), dtype)).cuda()
        got[i].append(grp.streams[i].resample_chunk(pre).cpu().numpy())
        want[i].append(solo[i].resample_chunk(pre).cpu().numpy())
    for r in range(rounds):
        x = torch.from_numpy(_sig(rng, (n,) + shape(chunk), dtype)).cuda()
        y, counts = grp.resample_chunks(x)
        assert y.shape[0] == n and len(counts) == n
        for i in range(n):
            got[i].append(y[i, :counts[i]].cpu().numpy())
            want[i].append(solo[i].resample_chunk(x[i]).cpu().numpy())
            assert np.array_equal(got[i][-1], want[i][-1]), (r, i)
        if r == 5:                                           # a member used alone between two group calls
            extra = torch.from_numpy(_sig(rng, shape(97), dtype)).cuda()
            got[3].append(grp.streams[3].resample_chunk(extra).cpu().numpy())
            want[3].append(solo[3].resample_chunk(extra).cpu().numpy())
    assert len({len(np.concatenate(g)) for g in got}) > 1    # the streams really are at different positions
    tail = torch.from_numpy(_sig(rng, shape(10), dtype)).cuda()
    for i in (0, 3, n - 1):
        assert np.array_equal(grp.streams[i].resample_chunk(tail, last=True).cpu().numpy(), solo[i].resample_chunk(tail, last=True).cpu().numpy())
    if np.issubdtype(dtype, np.integer):
        assert [s.num_clips() for s in grp.streams] == [s.num_clips() for s in solo]


def test_stream_group_falls_back_where_one_launch_does_not_apply(soxr):
    """Chunks of >= 4096 outputs per stream, or thousands of calls (ring compaction inside the group call): same frames."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(33)
    n = 5
    grp = dev.TensorStreamGroup(n, 48000, 44100, 1, dtype=torch.float32, quality="HQ")
    x = torch.from_numpy(_sig(rng, (n, 9000), np.float32)).cuda()      # 8268 outputs each: not the small-launch kernel
    y, counts = grp.resample_chunks(x)
    for i in range(n):
        ts = dev.TensorStream(48000, 44100, 1, dtype=torch.float32, quality="HQ")
        assert np.array_equal(y[i, :counts[i]].cpu().numpy(), ts.resample_chunk(x[i]).cpu().numpy())
    grp = dev.TensorStreamGroup(n, 44100, 16000, 1, dtype=torch.float32, quality="VHQ")
    sig = _sig(rng, (n, 441 * 600), np.float32)
    xd = torch.from_numpy(sig).cuda()
    parts = [[] for _ in range(n)]
    for a in range(0, sig.shape[1], 441):
        y, counts = grp.resample_chunks(xd[:, a:a + 441])
        for i in range(n):
            parts[i].append(y[i, :counts[i]])
    for i in range(n):
        tail = grp.streams[i].resample_chunk(xd[i, :0], last=True)
        assert np.array_equal(torch.cat(parts[i] + [tail]).cpu().numpy(), soxr.resample(sig[i], 44100, 16000, quality="VHQ"))


def test_stream_group_zero_frame_tick_is_not_a_flush(soxr):
    """A tick without input (x.shape[1] == 0: an empty device tensor's data_ptr() is NULL) must not be read as "end of input"
    by the C entry: the streams stay open, later ticks go on, and the frames are those of streams that never saw the empty
    tick (advisor, round 5).  The single-stream path is held to the same."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(35)
    n = 6
    grp = dev.TensorStreamGroup(n, 44100, 16000, 1, dtype=torch.int16, quality="HQ", dither_seeds=list(range(n)))
    solo = [dev.TensorStream(44100, 16000, 1, dtype=torch.int16, quality="HQ", dither_seed=i) for i in range(n)]
    sig = torch.from_numpy(_sig(rng, (n, 441 * 6), np.int16)).cuda()
    got = [[] for _ in range(n)]
    for r in range(6):
        if r in (0, 3):                                      # empty ticks: first call ever, and one in mid-stream
            y, counts = grp.resample_chunks(sig[:, :0])
            assert y.shape[0] == n and not any(s._ended for s in grp.streams)
            for i in range(n):
                got[i].append(y[i, :counts[i]])
            e1 = grp.streams[1].resample_chunk(sig[1, :0])   # ... and the single-stream form of the same
            assert e1.numel() == 0 and not grp.streams[1]._ended
        y, counts = grp.resample_chunks(sig[:, r * 441:(r + 1) * 441])
        for i in range(n):
            got[i].append(y[i, :counts[i]])
    for i in range(n):
        tail = grp.streams[i].resample_chunk(sig[i, :0], last=True)
        want = torch.cat([solo[i].resample_chunk(sig[i, r * 441:(r + 1) * 441], last=(r == 5)) for r in range(6)])
        assert np.array_equal(torch.cat(got[i] + [tail]).cpu().numpy(), want.cpu().numpy()), i


def test_streams_entry_through_the_c_abi_mixed_and_repeated_handles(soxr):
    """`hipsoxr_streams_process_device` called the way a C client would (pointer arrays; ragged chunk lengths per handle, one
    handle listed TWICE, one of another plan): handles the shared launch cannot take are processed one by one, in index order —
    every handle's frames equal those of single calls."""
    import ctypes as C
    import torch
    from soxr_amd import _native as nat, device as dev
    rng = np.random.default_rng(41)
    mk = lambda a, b: dev.TensorStream(a, b, 1, dtype=torch.float32, quality="HQ")                     # noqa: E731
    grp = [mk(48000, 44100), mk(48000, 44100), mk(48000, 44100), mk(44100, 16000)]
    ref = [mk(48000, 44100), mk(48000, 44100), mk(48000, 44100), mk(44100, 16000)]
    st = torch.cuda.current_stream().cuda_stream
    for rnd, order in enumerate(([0, 1, 2], [0, 1, 2, 3], [0, 1, 1, 2])):                               # same plan / mixed plans / a repeat
        n = len(order)
        lens = [int(rng.integers(100, 900)) for _ in order]
        xs = [torch.from_numpy(_sig(rng, (l,), np.float32)).cuda() for l in lens]
        outs = [torch.empty(l + 400, dtype=torch.float32, device="cuda") for l in lens]
        h = (C.c_void_p * n)(*[grp[i]._h.value for i in order])
        ins = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        ops = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        il = (C.c_size_t * n)(*lens)
        ol = (C.c_size_t * n)(*[o.shape[0] for o in outs])
        od = (C.c_size_t * n)()
        nat.check(nat.lib.hipsoxr_streams_process_device(h, n, ins, il, ops, ol, od, st))
        for k, i in enumerate(order):
            want = ref[i].resample_chunk(xs[k]).cpu().numpy()
            assert od[k] == len(want) and np.array_equal(outs[k][:od[k]].cpu().numpy(), want), (rnd, k)
