"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the product's multi-GPU module
(`soxr_amd.dist`: `shard`, `broadcast_bank`, `rank_info` — what bench.py and a one-process-per-GPU job call) —
contiguous clip sharding with no data-path collective, plus the one collective the path has (broadcast of the
shared filter bank from rank 0).  The per-clip arithmetic is stood in for by the oracle (there is no GPU here);
what is under test is the partition / broadcast logic."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from soxr_amd import device as dev, dist as sdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = dev.Plan(48000, 44100, "HQ")
    if rank != 0:                       # prove the broadcast is what installs the bank
        plan.set_bank(np.zeros((plan.L, plan.taps)))
    sdist.broadcast_bank(plan)               # default group, backend gloo: the bank travels as a host tensor
    bank = plan.bank()
    info = sdist.rank_info(plan)
    assert info["ranks_seen"] == world and info["banks_identical"] and info["backend"] == "gloo"
    lo, hi = sdist.shard(n_clips, world, rank)
    opl = oracle.plan(48000, 44100, "HQ")
    outs = {}
    for clip in range(lo, hi):          # each rank resamples only its own clips
        x = (np.random.default_rng(100 + clip).standard_normal(2000) * 0.25).astype(np.float32)
        outs[clip] = oracle.resample_channel(opl, x, "port_f32", bank=bank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), bank=bank, lo=lo, hi=hi,
             **{f"clip{c}": v for c, v in outs.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [7, 8])
def test_two_rank_sharding_and_bank_broadcast(tmp_path, n_clips, oracle):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_clips, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(world)]
    # both ranks hold rank 0's bank, bit for bit, and it is the designed bank (the product's own,
    # which in turn equals the oracle's independent numpy design to 1e-13: test_design_independent.py)
    assert np.array_equal(r[0]["bank"], r[1]["bank"])
    assert np.array_equal(r[0]["bank"], oracle.plan(48000, 44100, "HQ").port_bank)
    assert np.abs(r[0]["bank"] - oracle.plan(48000, 44100, "HQ").bank).max() <= 1e-13
    # shards are disjoint, contiguous and cover every clip
    assert int(r[0]["lo"]) == 0 and int(r[0]["hi"]) == int(r[1]["lo"]) and int(r[1]["hi"]) == n_clips
    # the union of per-rank results equals the unsharded computation
    opl = oracle.plan(48000, 44100, "HQ")
    for clip in range(n_clips):
        x = (np.random.default_rng(100 + clip).standard_normal(2000) * 0.25).astype(np.float32)
        want = oracle.resample_channel(opl, x, "port_f32")
        owner = 0 if clip < int(r[0]["hi"]) else 1
        assert np.array_equal(r[owner][f"clip{clip}"], want)


def test_shard_partition_properties():
    from soxr_amd import dist as sdist
    for n in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 4, 8):
            parts = [sdist.shard(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert sdist.shard(1024, 8, 3) == (384, 512)
    with pytest.raises(ValueError):
        sdist.shard(8, 2, 2)
    # a ragged corpus is dealt by total frames: every clip exactly once, loads within one longest clip of each other
    rng = np.random.default_rng(4)
    for world in (1, 2, 3, 8):
        lengths = [int(v) for v in rng.integers(5, 16, size=101) * 48000]
        parts = sdist.shard_by_frames(lengths, world)
        assert len(parts) == world and sorted(i for p in parts for i in p) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lengths)
        assert parts == sdist.shard_by_frames(lengths, world)          # deterministic
    assert sdist.shard_by_frames([], 3) == [[], [], []]
    sys.path.insert(0, ROOT)
    import bench                                 # bench.py uses the product's rule, not a copy of it
    assert bench.shard(1024, 8, 3) == sdist.shard(1024, 8, 3)
