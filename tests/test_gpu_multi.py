"""Tests that switch themselves ON when the box shows two or more GPUs (they are collected and skipped on one).

The reference's scaling model is threads over independent calls (reference tests/gil_bench.py:22-56); here the same
clips go over the devices of one node (`soxr_amd.dist.resample_batch`, one host thread per device) or over one process
per GPU (RCCL).  What has never run on one-GPU boxes and must hold the day an 8-GPU box runs the suite:
  * `resample_batch` over ALL devices == the same call on device 0: bit for bit on the exact engine, <= 1e-6 on AUTO;
  * ordering against the CALLER's current stream when workers run in other threads (round-4 advisor finding);
  * a real >= 2-rank RCCL bank broadcast, through torch and through `hipsoxr_plan_broadcast` — ranks != 0 start from a
    zeroed bank;
  * `python bench.py --gpus 2` with no launcher on the command line starts its own two ranks and says so in the line.
The partition rule itself (`shard_by_frames`) and the self-spawning bench harness are checked on any box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_dev():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


need2 = pytest.mark.skipif(_n_dev() < 2, reason="needs >= 2 HIP devices (this box shows %d)" % _n_dev())


def test_shard_by_frames_loads_within_one_clip():
    """Greedy longest-first: no device carries more than the lightest one plus one (longest) clip; every clip once."""
    from soxr_amd import dist as sdist
    rng = np.random.default_rng(12)
    for world in (2, 3, 8):
        for _ in range(20):
            lens = rng.integers(1, 15 * 48000, size=int(rng.integers(world, 200))).tolist()
            parts = sdist.shard_by_frames(lens, world)
            assert sorted(i for p in parts for i in p) == list(range(len(lens)))
            loads = [sum(lens[i] for i in p) for p in parts]
            assert max(loads) - min(loads) <= max(lens)
    assert sdist.shard_by_frames([], 4) == [[], [], [], []]


@pytest.mark.gpu
@need2
def test_resample_batch_over_all_devices_equals_one_device():
    import torch
    from soxr_amd import device as dev, dist as sdist
    nd = torch.cuda.device_count()
    rng = np.random.default_rng(21)
    lens = rng.integers(20000, 200000, size=4 * nd + 3)
    host = [(rng.standard_normal(int(n)) * 0.25).astype(np.float32) for n in lens]
    # a mix of host arrays and tensors living on every device
    clips = [c if i % 3 == 0 else torch.from_numpy(c).to(f"cuda:{i % nd}") for i, c in enumerate(host)]
    for kernel, exact in ((dev.KERNEL_EXACT, True), (0, False)):
        one = sdist.resample_batch(host, 48000, 44100, "VHQ", devices=[0], kernel=kernel)
        many = sdist.resample_batch(clips, 48000, 44100, "VHQ", kernel=kernel)
        for d in range(nd):
            torch.cuda.synchronize(d)
        used = {m.device.index for m in many if not isinstance(m, np.ndarray)}
        assert len(used) >= 2, used          # the work really was spread
        for a, b in zip(one, many):
            b = b if isinstance(b, np.ndarray) else b.cpu().numpy()
            assert a.shape == b.shape
            if exact:
                assert np.array_equal(a, b)
            else:
                assert np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) <= 1e-6 * np.sqrt(np.mean(a.astype(np.float64) ** 2))


@pytest.mark.gpu
@need2
def test_resample_batch_orders_against_the_callers_stream():
    """Inputs produced on a NON-default stream, results consumed on it, no host synchronisation in between: the worker
    threads must order their side streams against the caller's streams (thread-local `current_stream` is not it)."""
    import torch
    from soxr_amd import device as dev, dist as sdist
    nd = torch.cuda.device_count()
    streams = [torch.cuda.Stream(device=d) for d in range(nd)]
    ref_in = [torch.randn(400000, device=f"cuda:{d}") * 0.25 for d in range(nd)]
    for d in range(nd):
        torch.cuda.synchronize(d)
    want = sdist.resample_batch([r.clone() for r in ref_in], 48000, 44100, "VHQ", kernel=dev.KERNEL_EXACT)
    for d in range(nd):
        torch.cuda.synchronize(d)
    for _ in range(5):
        clips, sums = [], []
        import contextlib
        with contextlib.ExitStack() as es:
            for d in range(nd):
                es.enter_context(torch.cuda.stream(streams[d]))
            for d in range(nd):
                with torch.cuda.device(d):
                    big = torch.empty(64 << 20, device=f"cuda:{d}").normal_()   # keeps stream d busy in front of the clip
                    clips.append(ref_in[d] * (big[0] * 0 + 1))                    # produced on stream d, behind `big`
            outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", kernel=dev.KERNEL_EXACT)
            for o in outs:
                with torch.cuda.device(o.device):
                    sums.append(o.clone())                                        # consumed on the caller's stream
        for d in range(nd):
            torch.cuda.synchronize(d)
        for w, g in zip(want, sums):
            assert torch.equal(w.cpu(), g.cpu())


@pytest.mark.gpu
@need2
def test_two_rank_rccl_bank_broadcast():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "tests", "_multi_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "MULTI_OK 2" in p.stdout


@pytest.mark.gpu
@need2
def test_bench_gpus_2_runs_two_rccl_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--no-batch", "--no-cpu"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2 and d["ranks"]["backend"] == "nccl (RCCL)" and d["ranks"]["banks_identical"]
    assert len({r["device"] for r in d["ranks"]["ranks"]}) == 2


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_ranks_gloo_harness():
    """`python bench.py --gpus 2` — no torchrun on the command line — re-executes itself as two ranks (here sharing this
    box's GPU over gloo: BENCH_DIST_BACKEND=gloo is the harness switch; real runs use RCCL, one rank per GPU)."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-batch", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2 and d["ranks"]["backend"] == "gloo" and d["ranks"]["banks_identical"]


@pytest.mark.gpu
def test_bench_gpus_8_gloo_harness_partitions_the_batch():
    """The driver's largest line, `--gpus 8`, proven before hardware runs it: eight self-started ranks (sharing this box's GPU
    over gloo), every rank seen, one bank, and the strong-scaling batch partitioned [0, 128) ... [896, 1024) with no clip
    dropped or held twice.  (The scaling model is the reference's threads over independent calls,
    /root/reference/tests/gil_bench.py:22-56: nothing is exchanged but the bank.)"""
    import time
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--kernels-only", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["ranks"]["ranks_seen"] == 8 and len(d["ranks"]["ranks"]) == 8 and d["ranks"]["backend"] == "gloo" and d["ranks"]["banks_identical"]
    assert sorted(r["rank"] for r in d["ranks"]["ranks"]) == list(range(8))
    assert d["batch_strong"]["shards"] == [[128 * r, 128 * (r + 1)] for r in range(8)]
    assert wall < 120, wall


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_the_box_has():
    n = _n_dev() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env={k: v for k, v in os.environ.items() if k != "BENCH_DIST_BACKEND"})
    assert p.returncode != 0 and "HIP device" in (p.stderr + p.stdout)
