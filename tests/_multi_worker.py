"""Worker of tests/test_gpu_multi.py: one rank of a real multi-GPU job (one process per GPU, backend nccl = RCCL).

Rank != 0 starts from a ZEROED bank, so that an output equal to rank 0's can only come from the broadcast:
  1. `soxr_amd.dist.broadcast_bank` through torch's communicator;
  2. `hipsoxr_plan_broadcast` (C ABI) over a raw ncclComm_t made with RCCL's own C API (unique id shared through torch).
Every rank then resamples the same seeded clip with the exact engine and the ranks compare SHA-256 digests."""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-soxr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from soxr_amd import _native as nat, device as dev, dist as sdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    x = (np.random.default_rng(3).standard_normal(96000) * 0.25).astype(np.float32)
    xt = torch.from_numpy(x).to(device)

    def digest(plan):
        y = dev.resample_tensor(plan, xt, kernel=dev.KERNEL_EXACT)
        torch.cuda.synchronize(device)
        return hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()

    def zeroed_plan():
        p = dev.Plan(48000, 44100, "VHQ")
        if rank != 0:
            p.set_bank(np.zeros_like(p.bank()))
            assert not p.bank().any()
        return p

    # 1. through torch.distributed (RCCL)
    p1 = zeroed_plan()
    sdist.broadcast_bank(p1, device=device)
    info = sdist.rank_info(p1, device=device)
    assert info["ranks_seen"] == world and info["backend"] == "nccl (RCCL)" and info["banks_identical"], info
    assert len({r["device"] for r in info["ranks"]}) == world, info
    d1 = [None] * world
    dist.all_gather_object(d1, digest(p1))
    assert len(set(d1)) == 1, d1

    # 2. the native entry on a raw communicator
    rccl = None
    for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"):
        try:
            rccl = C.CDLL(cand, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None, "no RCCL library to load"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    box = [bytes(uid) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    C.memmove(C.byref(uid), box[0], 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    try:
        p2 = zeroed_plan()
        st = torch.cuda.current_stream(device).cuda_stream
        nat.check(nat.lib.hipsoxr_plan_broadcast(p2.handle, comm, 0, rank, st))
        d2 = [None] * world
        dist.all_gather_object(d2, (digest(p2), sdist.bank_digest(p2)))
        assert len(set(d2)) == 1, d2
        assert d2[0][0] == d1[0]
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("MULTI_OK", world, d1[0][:16])


if __name__ == "__main__":
    main()
