"""The native multi-GPU entry: hipsoxr_plan_broadcast over an RCCL communicator (no torch in the path).

One GPU is all a gpurun box has, so what can be checked here is a one-rank communicator made with RCCL's own
C API through ctypes (ncclGetUniqueId / ncclCommInitRank): the call resolves ncclBroadcast in the process,
runs on the given stream and leaves the root's bank intact and usable.  The N > 1 semantics (ranks != root
install what they receive) share the code path below the broadcast with hipsoxr_plan_set_bank, which
tests/test_dist_gloo.py covers with two processes.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rccl():
    import torch
    for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"):
        try:
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("no RCCL library to load")


def test_plan_broadcast_single_rank_communicator(oracle):
    import torch
    from soxr_amd import _native as nat, device as dev
    torch.zeros(1, device="cuda")
    rccl = _rccl()

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        plan = dev.Plan(48000, 44100, "VHQ")
        before = plan.bank().copy()
        st = torch.cuda.current_stream().cuda_stream
        nat.check(nat.lib.hipsoxr_plan_broadcast(plan.handle, comm, 0, 0, st))
        assert np.array_equal(plan.bank(), before)
        x = torch.randn(48000, device="cuda") * 0.25
        y = dev.resample_tensor(plan, x, kernel=dev.KERNEL_EXACT).cpu().numpy()
        assert np.array_equal(y, oracle.resample(x.cpu().numpy(), 48000, 44100, "VHQ", mode="port"))
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
