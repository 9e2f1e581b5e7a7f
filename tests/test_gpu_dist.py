"""soxr_amd.dist on the GPU: ragged batches as one job (hipsoxr_job_t::clip_table), the one-process / one-thread-per-
device `resample_batch`, and BASELINE configs[3] WHOLE on one GPU (1024 x 10 s clips — the strong-scaling anchor)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXACT, FFT, FFT_F64 = 6, 5, 8


def _rms(a):
    return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))


def test_resample_batch_three_unequal_clips_bit_exact(oracle):
    """Three clips of unequal length, one call, canonical-order engine: each result equals the oracle's port of that
    clip alone, bit for bit (ragged jobs outside the frequency-domain engine are served clip by clip)."""
    from soxr_amd import dist as sdist, device as dev
    rng = np.random.default_rng(31)
    clips = [(rng.standard_normal(n) * 0.25).astype(np.float32) for n in (48000, 1234, 100001)]
    outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], kernel=dev.KERNEL_EXACT)
    assert len(outs) == 3
    for x, y in zip(clips, outs):
        want = oracle.resample(x, 48000, 44100, "VHQ", mode="port")
        assert isinstance(y, np.ndarray) and y.dtype == np.float32 and y.shape == want.shape
        assert np.array_equal(y, want)
    # the same through integer I/O and two channels (clip-by-clip path), incl. an empty and a one-frame clip
    ci = [(rng.standard_normal((n, 2)) * 4000).astype(np.int16) for n in (5000, 0, 1, 20011)]
    oi = sdist.resample_batch(ci, 44100, 16000, "HQ", devices=[0], kernel=dev.KERNEL_EXACT)
    for x, y in zip(ci, oi):
        want = oracle.resample(x, 44100, 16000, "HQ")
        assert y.shape == want.shape and np.array_equal(y, want)


@pytest.mark.parametrize("dtype,kernel,tol", [(np.float32, 0, 1e-6), (np.float32, FFT_F64, 5e-8), (np.float64, 0, 2e-9)])
def test_ragged_job_is_one_launch_of_the_fft_engine(oracle, dtype, kernel, tol):
    """Unit-stride float clips of unequal length: ONE launch of k_fft_pair2 (every workgroup reads its clip's row of
    the table), results within the engine's bar of the oracle's float64 direct form — first and last outputs of every
    clip included (the zero extension past each clip's own end is the hardware range check on ITS length)."""
    import torch
    from soxr_amd import dist as sdist, device as dev
    rng = np.random.default_rng(32)
    lens = [9000, 100000, 1, 52345, 300007, 4704, 0, 77777]
    clips = [torch.from_numpy((rng.standard_normal(n) * 0.25).astype(dtype)).cuda() for n in lens]
    plan = dev.Plan(48000, 44100, "VHQ")
    job = sdist.RaggedJob(plan, clips, kernel=kernel)
    job.y.fill_(7.0)                                            # anything not written would show
    job.launch()
    torch.cuda.synchronize()
    outs = job.outputs()
    for n, x, y in zip(lens, clips, outs):
        ref = oracle.resample(x.cpu().numpy().astype(np.float64), 48000, 44100, "VHQ", mode="ref")
        y = y.cpu().numpy().astype(np.float64)
        assert y.shape == ref.shape, n
        if len(ref):
            assert _rms(y - ref) <= tol * max(_rms(ref), 0.05), (n, _rms(y - ref) / max(_rms(ref), 0.05))
            # the head and the tail on their own
            w = min(len(ref), 300)
            assert _rms(y[:w] - ref[:w]) <= 4 * tol * 0.25 and _rms(y[-w:] - ref[-w:]) <= 4 * tol * 0.25, n
    # and close to what each clip gives as a job of its own (a lone clip may be cut into smaller blocks: not bit-equal)
    for x, y in zip(clips, outs):
        if x.numel() >= 8192 * 2:
            alone = dev.resample_tensor(plan, x, kernel=kernel)
            assert _rms((alone - y).cpu().numpy()) <= 2 * tol * 0.25


def test_ragged_job_rejects_bad_tables():
    import torch
    from soxr_amd import dist as sdist, device as dev, _native as nat
    plan = dev.Plan(48000, 44100, "HQ")
    clips = [torch.zeros(5000, device="cuda"), torch.zeros(7000, device="cuda")]
    job = sdist.RaggedJob(plan, clips)
    job._table[1, 3] += 1                                       # more output than the plan allows for this clip
    with pytest.raises(RuntimeError):
        job.launch()
    job._table[1, 3] -= 1
    job._table[0, 0] = -4                                       # an offset in front of the job's base
    with pytest.raises(RuntimeError):
        job.launch()
    job._table[0, 0] = 0
    ref = [o.clone() for o in (job.launch(), job.outputs())[1]]
    job._job.clip_table_dev = None                              # host table only: the library uploads it in stream order
    job.y.zero_()
    job.launch()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, job.outputs()))
    with pytest.raises(ValueError):
        sdist.RaggedJob(plan, [torch.zeros(10, device="cuda"), torch.zeros((10, 2), device="cuda")])


def test_resample_batch_threads_over_devices_and_order():
    """One thread + stream per device (here: the one GPU listed twice — the code path of an 8-GPU node; results are
    in the caller's order whatever the partition)."""
    import torch
    from soxr_amd import dist as sdist, device as dev
    rng = np.random.default_rng(33)
    clips = [(rng.standard_normal(20000 + 997 * i) * 0.25).astype(np.float32) for i in range(7)]
    one = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], kernel=dev.KERNEL_EXACT)
    two = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0, 0], kernel=dev.KERNEL_EXACT)
    three = sdist.resample_batch([torch.from_numpy(c) for c in clips], 48000, 44100, "VHQ", devices=[0, 0, 0], kernel=dev.KERNEL_EXACT)
    for a, b, c in zip(one, two, three):
        assert np.array_equal(a, b) and torch.is_tensor(c) and c.is_cuda and np.array_equal(a, c.cpu().numpy())
    assert sdist.resample_batch([], 48000, 44100) == []


def test_device_clips_are_resampled_in_place_and_in_stream_order():
    """Device tensors are not packed into a second buffer (the job table addresses them where they lie), and work queued
    on the caller's current stream just before the call — the producer of the clips — is seen by the side stream that
    runs the job; the caller's stream in turn sees finished results without a host synchronisation."""
    import torch
    from soxr_amd import dist as sdist, device as dev
    plan = dev.Plan(48000, 44100, "VHQ")
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    base = [torch.randn(300000 + 1111 * i, device="cuda", generator=g) * 0.1 for i in range(5)]
    want = [dev.resample_tensor(plan, b * 2.0 + 0.25, kernel=dev.KERNEL_EXACT) for b in base]
    torch.cuda.synchronize()
    for trial in range(3):
        big = torch.randn(1 << 26, device="cuda", generator=g)
        for _ in range(4):
            big = big * 1.0001 + 1e-3          # a few ms of work in front of the producers on the current stream
        clips = [b * 2.0 + 0.25 for b in base]  # produced on the current stream, not yet finished
        outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], kernel=dev.KERNEL_EXACT)
        acc = [o + 0.0 for o in outs]           # consumed on the current stream, no synchronize in between
        del clips
        torch.cuda.synchronize()
        for a, w in zip(acc, want):
            assert torch.equal(a, w), trial
    job = sdist.RaggedJob(plan, base)
    assert job._job.in_ == min(b.data_ptr() for b in base)      # no packed copy of the input


def test_host_corpus_goes_through_the_staging_pipeline(oracle):
    """numpy clips of unequal length, several blocks of the pinned staging ring (small block size forced), float32 and
    int16: every clip equals the one-shot host surface bit for bit (canonical-order engine)."""
    import soxr_amd as soxr
    from soxr_amd import dist as sdist, device as dev
    rng = np.random.default_rng(77)
    clips = [(rng.standard_normal(30000 + 4001 * (i % 7)) * 0.25).astype(np.float32) for i in range(23)]
    for pinned in (None, True, False):      # results as views of the pinned D2H buffers (the default at this size) / copied out
        outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0, 0], kernel=dev.KERNEL_EXACT, block_bytes=400000,
                                    pinned_results=pinned)
        for c, o in zip(clips, outs):
            assert isinstance(o, np.ndarray) and o.flags.writeable and np.array_equal(o, soxr.resample(c, 48000, 44100, quality="VHQ"))
    # pinned results own their buffers: a second batch through the same pipes must not touch the first one's arrays
    outs_pinned = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], kernel=dev.KERNEL_EXACT, block_bytes=400000, pinned_results=True)
    keep = [o.copy() for o in outs_pinned]
    sdist.resample_batch([c[::-1].copy() for c in clips], 48000, 44100, "VHQ", devices=[0], kernel=dev.KERNEL_EXACT, block_bytes=400000,
                         pinned_results=True)
    for a, b in zip(outs_pinned, keep):
        assert np.array_equal(a, b)
    outs_pinned[0][:] = 0                   # ... and are ordinary writable arrays
    ci = [(rng.standard_normal((9000 + 501 * i, 2)) * 5000).astype(np.int16) for i in range(9)]
    oi = sdist.resample_batch(ci, 44100, 16000, "HQ", devices=[0], kernel=dev.KERNEL_EXACT, block_bytes=100000)
    for c, o in zip(ci, oi):
        assert np.array_equal(o, soxr.resample(c, 44100, 16000, quality="HQ"))


def test_config3_whole_batch_on_one_gpu(oracle):
    """BASELINE configs[3] whole: 1024 independent 10 s clips, VHQ 48k -> 44.1k, float32, on ONE GPU (3.8 GB of
    signal; the N = 1 point of the strong-scaling line).  Clip independence (batched == alone, bit for bit, for the
    same engine) and the 1e-6 bar against the oracle's float64 direct form on windows of three clips."""
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(48000, 44100, "VHQ")
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn((1024, 480000, 1), device="cuda", generator=g) * 0.25
    y = dev.resample_tensor(plan, x)
    assert tuple(y.shape) == (1024, 441000, 1)
    rng = np.random.default_rng(40)
    opl = oracle.plan(48000, 44100, "VHQ")
    for clip in (0, 511, 1023):
        alone = dev.resample_tensor(plan, x[clip, :, 0].contiguous().repeat(128, 1)[:, :, None])  # a 128-clip job: same block size
        assert torch.equal(alone[5, :, 0], y[clip, :, 0])
        xc, yc = x[clip, :, 0].cpu().numpy().astype(np.float64), y[clip, :, 0].cpu().numpy().astype(np.float64)
        for k0 in (0, 441000 - 2000, int(rng.integers(0, 439000))):
            want = oracle.resample_channel(opl, xc, "ref", k0=k0, n_out=2000)
            assert _rms(yc[k0:k0 + 2000] - want) <= 1e-6 * max(_rms(want), 0.05)
    del x, y
    torch.cuda.empty_cache()


def test_host_pipeline_surfaces_errors_instead_of_hanging(monkeypatch):
    """Fault injection into `_HostPipe` (three host threads, three streams): an error in the issuing loop mid-corpus — the
    library refusing the third block — and an error inside a copy thread must both come back as exceptions from
    `resample_batch`, with every helper thread stopped, and the pipe must serve the next call normally."""
    import threading
    from soxr_amd import dist as sdist, _native as nat
    rng = np.random.default_rng(4)
    clips = [(rng.standard_normal(48000) * 0.25).astype(np.float32) for _ in range(24)]
    kw = dict(devices=[0], block_bytes=4 * 48000 * 3, kernel=6)          # 8 blocks of 3 clips
    good = sdist.resample_batch(clips, 48000, 44100, "VHQ", **kw)
    before = threading.active_count()

    # (a) the C entry fails on its third call
    real = nat.lib.hipsoxr_run_device
    calls = {"n": 0}

    class Failing:
        def __call__(self, *a):
            calls["n"] += 1
            return b"injected failure" if calls["n"] == 3 else real(*a)
    monkeypatch.setattr(nat.lib, "hipsoxr_run_device", Failing())
    done = {}

    def run():
        try:
            sdist.resample_batch(clips, 48000, 44100, "VHQ", **kw)
            done["r"] = "no error"
        except Exception as e:  # noqa: BLE001
            done["r"] = e
    t = threading.Thread(target=run); t.start(); t.join(60)
    assert not t.is_alive(), "resample_batch hangs after a failed launch"
    assert isinstance(done["r"], RuntimeError) and "injected failure" in str(done["r"]), done
    monkeypatch.setattr(nat.lib, "hipsoxr_run_device", real)

    # (b) a copy thread fails (pageable results: the unstager allocates)
    real_empty = np.empty
    hits = {"n": 0}

    def bad_empty(*a, **k):
        if threading.current_thread() is not threading.main_thread() and a and a[0] == (44100,):
            hits["n"] += 1
            if hits["n"] == 5:
                raise MemoryError("injected allocation failure")
        return real_empty(*a, **k)
    monkeypatch.setattr(np, "empty", bad_empty)
    t = threading.Thread(target=lambda: done.__setitem__("r2", _try(lambda: sdist.resample_batch(clips, 48000, 44100, "VHQ", pinned_results=False, **kw))))
    t.start(); t.join(60)
    assert not t.is_alive(), "resample_batch hangs after a failed copy thread"
    assert isinstance(done["r2"], MemoryError), done
    monkeypatch.setattr(np, "empty", real_empty)

    # the pipes still work, and no helper thread is left behind
    again = sdist.resample_batch(clips, 48000, 44100, "VHQ", **kw)
    assert all(np.array_equal(a, b) for a, b in zip(good, again))
    assert threading.active_count() <= before


def _try(f):
    try:
        return f()
    except Exception as e:  # noqa: BLE001
        return e


def test_host_pipeline_concurrent_callers_do_not_share_slots():
    """Two Python threads in `resample_batch` on host arrays at once (the module's concurrency model): each run checks a
    pipe out for itself — results equal the single-threaded ones bit for bit."""
    import threading
    from soxr_amd import dist as sdist
    rng = np.random.default_rng(6)
    sets = [[(rng.standard_normal(int(n)) * 0.25).astype(np.float32) for n in rng.integers(20000, 90000, size=40)] for _ in range(3)]
    want = [sdist.resample_batch(c, 48000, 44100, "VHQ", devices=[0], kernel=6, block_bytes=1 << 20) for c in sets]
    got = [None] * 3

    def run(i):
        for _ in range(3):
            got[i] = sdist.resample_batch(sets[i], 48000, 44100, "VHQ", devices=[0], kernel=6, block_bytes=1 << 20)
    ts = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    [t.start() for t in ts]; [t.join(300) for t in ts]
    for w, g in zip(want, got):
        assert all(np.array_equal(a, b) for a, b in zip(w, g))


def test_zero_length_host_share_is_served():
    """A device's share made of zero-length host clips (round-4 advisor finding): empty results, no dead stager."""
    from soxr_amd import dist as sdist
    outs = sdist.resample_batch([np.zeros(0, np.float32), np.zeros((0, 2), np.float32)[:, 0].copy()], 48000, 44100, "VHQ", devices=[0])
    assert [o.shape for o in outs] == [(0,), (0,)]
    outs = sdist.resample_batch([np.zeros(0, np.float32), np.ones(4800, np.float32)], 48000, 44100, "VHQ", devices=[0])
    assert outs[0].shape == (0,) and outs[1].shape == (4410,)
