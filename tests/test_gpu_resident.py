"""Resident kernel (extension, HIPSOXR_RESIDENT / ResampleStream(resident=True)): synchronous small chunks are
served by a kernel that stays on the GPU and is fed through a mailbox in pinned memory
(python-soxr_amd/csrc/kernels.hip k_chain_resident, engine.cpp resident_emit).  The contract is the synchronous
stream's, bit for bit and call for call (reference: CSoxr::process, /root/reference/src/soxr_ext.cpp:162-187)."""
import time

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module")
def soxr():
    import soxr_amd
    return soxr_amd


def _signal(dtype, n, ch, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, ch)) if ch > 1 else rng.standard_normal(n)
    return (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)


def _run(rs, x, chunks, pause=0.0):
    parts, a, i = [], 0, 0
    while a < len(x):
        c = chunks[i % len(chunks)]
        parts.append(rs.resample_chunk(x[a:a + c], last=(a + c >= len(x))))
        a += c; i += 1
        if pause:
            time.sleep(pause)
    return parts


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
@pytest.mark.parametrize("rates,quality,ch,chunk", [((44100, 16000), "VHQ", 1, 441), ((48000, 44100), "HQ", 2, 480),
                                                     ((44100, 48000), "VHQ", 1, 100), ((8000, 22050), "MQ", 3, 17),
                                                     ((44100, 16000), "VHQ", 1, 4410)])
def test_resident_equals_synchronous(soxr, dtype, rates, quality, ch, chunk):
    x = _signal(dtype, 30000, ch, 41)
    ref = _run(soxr.ResampleStream(*rates, ch, dtype=dtype, quality=quality), x, [chunk])
    got = _run(soxr.ResampleStream(*rates, ch, dtype=dtype, quality=quality, resident=True), x, [chunk])
    assert [len(p) for p in got] == [len(p) for p in ref]          # the same frames in the same calls
    assert np.array_equal(np.concatenate(got), np.concatenate(ref))
    assert np.array_equal(np.concatenate(got), soxr.resample(x, *rates, quality=quality))


def test_resident_kernel_leaves_and_comes_back(soxr):
    """Calls spaced further apart than the idle time (1 ms): the instance leaves by itself, the next call starts
    another one; a device-wide synchronisation in between returns (it waits for the instance at most)."""
    import torch
    x = _signal(np.int16, 44100, 1, 42)
    ref = np.concatenate(_run(soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ"), x, [441]))
    rs = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ", resident=True)
    parts, t_sync = [], 0.0
    for i, a in enumerate(range(0, len(x), 441)):
        parts.append(rs.resample_chunk(x[a:a + 441], last=(a + 441 >= len(x))))
        if i % 10 == 3:
            time.sleep(0.004)
        if i % 10 == 7:
            t0 = time.perf_counter()
            torch.cuda.synchronize()
            t_sync = max(t_sync, time.perf_counter() - t0)
    assert np.array_equal(np.concatenate(parts), ref)
    assert t_sync < 0.5


def test_resident_mixed_chunks_clear_and_clip_count(soxr):
    """Chunk sizes from a few frames to beyond what the resident form serves (the ring then moves to device memory
    and the stream continues on the ordinary path), clear() and re-use, the clip counter."""
    x = _signal(np.int16, 150000, 2, 43)
    x[5000:5200] = 32767  # overshoot: clipped samples
    chunks = [441, 7, 2000, 441, 441, 1, 9000, 441, 50000, 441, 300]
    for rep in range(2):
        a = soxr.ResampleStream(44100, 16000, 2, dtype=np.int16, quality="HQ")
        b = soxr.ResampleStream(44100, 16000, 2, dtype=np.int16, quality="HQ", resident=True)
        for r in range(2):
            ya, yb = _run(a, x, chunks), _run(b, x, chunks)
            assert [len(p) for p in ya] == [len(p) for p in yb]
            assert np.array_equal(np.concatenate(ya), np.concatenate(yb))
            assert a.num_clips() == b.num_clips() and a.num_clips() > 0
            a.clear(); b.clear()
        chunks = chunks[::-1]


def test_resident_many_streams_at_once(soxr):
    """Several resident streams in one process, interleaved calls: each has its own mailbox and instance."""
    x = _signal(np.float32, 20000, 1, 44)
    want = np.concatenate(_run(soxr.ResampleStream(48000, 44100, 1, quality="HQ"), x, [480]))
    streams = [soxr.ResampleStream(48000, 44100, 1, quality="HQ", resident=True) for _ in range(6)]
    parts = [[] for _ in streams]
    for a in range(0, len(x), 480):
        for k, rs in enumerate(streams):
            parts[k].append(rs.resample_chunk(x[a:a + 480], last=(a + 480 >= len(x))))
    for p in parts:
        assert np.array_equal(np.concatenate(p), want)


def test_resident_budget(soxr):
    """More resident streams than the process admits resident workgroups for (512; a 480-frame-chunk stream holds
    ~70): the ones over the budget run on the ordinary path, every stream's output is the same."""
    x = _signal(np.float32, 6000, 1, 45)
    want = np.concatenate(_run(soxr.ResampleStream(48000, 44100, 1, quality="HQ"), x, [480]))
    streams = [soxr.ResampleStream(48000, 44100, 1, quality="HQ", resident=True) for _ in range(24)]
    parts = [[] for _ in streams]
    for a in range(0, len(x), 480):
        for k, rs in enumerate(streams):
            parts[k].append(rs.resample_chunk(x[a:a + 480], last=(a + 480 >= len(x))))
    for p in parts:
        assert np.array_equal(np.concatenate(p), want)


def test_environment_switch_runs_the_stream_contract():
    """HIPSOXR_RESIDENT in the environment turns every eligible stream of the process into a resident one (the way a
    user of the libsoxr-named library would switch it on): the reference's behavioural contract and the stream parity
    tests pass unchanged."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIPSOXR_RESIDENT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_reference_contract.py"),
                        os.path.join(root, "tests", "test_soxr_abi.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_resident_clear_then_longer_first_chunk(soxr):
    """clear() restarts the ring at frame 0; a first chunk LONGER than what the previous signal had left staged must
    not inherit its head from the device-side copy of the old ring."""
    a = _signal(np.float32, 3000, 1, 46)
    b = _signal(np.float32, 3000, 1, 47)
    rs = soxr.ResampleStream(48000, 44100, 1, quality="HQ", resident=True)
    ref = soxr.ResampleStream(48000, 44100, 1, quality="HQ")
    for x, chunks in ((a[:100], [100]), (b, [441, 300, 1000]), (a[:50], [50]), (a, [2000, 50])):
        got = np.concatenate(_run(rs, x, chunks))
        want = np.concatenate(_run(ref, x, chunks))
        assert np.array_equal(got, want)
        rs.clear(); ref.clear()


def test_small_chunk_stream_after_a_large_chunk_stream_still_gets_the_host_ring():
    """Finished streams hand their buffers to the next stream of the process (engine.cpp stream pool).  A stream fed
    96000-frame chunks leaves a DEVICE ring behind; the small-chunk stream after it must still get the pinned host
    ring — and with it the resident kernel — instead of inheriting the slow path for the rest of the process (round 3:
    441-frame calls ran 23 us instead of 15 after any large-chunk stream).  Checked through behaviour: results are
    bit-identical either way, so compare per-call time against a fresh small-chunk stream in the same process."""
    import time
    import soxr_amd as soxr
    rng = np.random.default_rng(21)
    x = (rng.standard_normal(44100 * 4) * 5000).astype(np.int16)

    def per_call(chunk):
        rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ")
        rs.resample_chunk(x[:chunk]); rs.clear()
        outs, n, t0 = [], 0, time.perf_counter()
        for a in range(0, len(x), chunk):
            outs.append(rs.resample_chunk(x[a:a + chunk], last=(a + chunk >= len(x)))); n += 1
        return (time.perf_counter() - t0) / n, np.concatenate(outs)

    # (best of three on either side: a timing test must not trip over one hiccup of the box)
    fresh = [per_call(441) for _ in range(3)]
    t_fresh, y_fresh = min(t for t, _ in fresh), fresh[0][1]
    after = []
    for _ in range(3):
        per_call(96000)                   # leaves a device ring in the pool
        after.append(per_call(441))
    t_after = min(t for t, _ in after)
    assert all(np.array_equal(y_fresh, y) for _, y in fresh + after)
    assert t_after < 1.35 * t_fresh + 2e-6, (t_fresh, t_after)


def test_default_streams_never_park_a_kernel_and_auto_ones_do_so_only_while_fed(soxr):
    """ADVICE (round 3): the path that turns resident by itself is OPT-IN now.  A stream created without a flag, fed small
    chunks back to back, leaves nothing spinning on the GPU: a device-wide synchronisation right behind it returns at once.
    With resident="auto" the same feed turns the path on (and the frames stay the synchronous stream's, call for call); while
    another thread keeps feeding, device-wide synchronisations from this thread still return (bounded by the idle time plus a
    relaunch, not for ever); once the feed stops — a gap in the run — the stream drops back by itself."""
    import threading
    import torch
    x = _signal(np.int16, 44100 * 2, 1, 43)
    ref = _run(soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ"), x, [441])
    # no flag: nothing resident
    rs = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ")
    waits = []
    for rep in range(5):
        for a in range(0, 441 * 40, 441):
            rs.resample_chunk(x[a:a + 441])
        t0 = time.perf_counter(); torch.cuda.synchronize(); waits.append(time.perf_counter() - t0)
    assert sorted(waits)[2] < 0.5e-3, waits              # (a resident instance holds a device sync for its 1 ms idle time)
    # opt-in: same frames in the same calls
    auto = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ", resident="auto")
    got = _run(auto, x, [441])
    assert [len(p) for p in got] == [len(p) for p in ref] and np.array_equal(np.concatenate(got), np.concatenate(ref))
    # concurrent device-wide synchronisations while another thread feeds an auto stream
    feeder_done, sync_times = threading.Event(), []
    auto2 = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ", resident="auto")
    out2 = []

    def feed():
        try:
            out2.extend(_run(auto2, x, [441]))
        finally:
            feeder_done.set()

    th = threading.Thread(target=feed); th.start()
    while not feeder_done.is_set():
        t0 = time.perf_counter(); torch.cuda.synchronize(); sync_times.append(time.perf_counter() - t0)
        time.sleep(0.002)
    th.join()
    assert np.array_equal(np.concatenate(out2), np.concatenate(ref))
    assert sync_times and max(sync_times) < 0.25, max(sync_times)    # never starved: every synchronisation came back
    # the feed has stopped: after a gap the stream has dropped back — a device sync behind one more small call is immediate
    time.sleep(0.01)
    auto3 = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ", resident="auto")
    for a in range(0, 441 * 40, 441):
        auto3.resample_chunk(x[a:a + 441])                           # (turned resident by now)
    time.sleep(0.01)                                                 # the run breaks
    auto3.resample_chunk(x[:441])                                    # ordinary path again
    t0 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert dt < 0.5e-3, dt


def test_watchdog_bounds_an_idle_resident_instance_whatever_the_environment_says():
    """HIPSOXR_RESIDENT_IDLE_US = 5 s in a process of its own: the instance still leaves after the 20 ms watchdog
    (csrc/device.h kResidentWatchdogUs), so a device-wide synchronisation behind an idle resident stream returns in
    tens of milliseconds, not seconds — and the stream carries on afterwards with the same frames."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, time, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import soxr_amd as soxr
rng = np.random.default_rng(5)
x = (rng.standard_normal(441 * 60) * 5000).astype(np.int16)
ref = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ")
rs = soxr.ResampleStream(44100, 16000, 1, dtype=np.int16, quality="VHQ", resident=True)
a = [rs.resample_chunk(x[i:i + 441]) for i in range(0, 441 * 30, 441)]
b = [ref.resample_chunk(x[i:i + 441]) for i in range(0, 441 * 30, 441)]
t0 = time.perf_counter(); torch.cuda.synchronize(); wait = time.perf_counter() - t0      # the instance is idle and spinning
a += [rs.resample_chunk(x[i:i + 441]) for i in range(441 * 30, 441 * 60, 441)]
b += [ref.resample_chunk(x[i:i + 441]) for i in range(441 * 30, 441 * 60, 441)]
print("WATCHDOG " + json.dumps({"wait": wait, "same": bool(np.array_equal(np.concatenate(a), np.concatenate(b)))}))
""" % (ROOT, os.path.join(ROOT, "python-soxr_amd"))
    env = dict(os.environ, HIPSOXR_RESIDENT_IDLE_US="5000000")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("WATCHDOG ")][-1][len("WATCHDOG "):])
    assert d["same"]
    assert d["wait"] < 0.2, d            # 20 ms watchdog (+ relaunch slack); 5 s if the environment were obeyed
