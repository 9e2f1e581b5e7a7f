"""Batches of independent clips over the GPUs of one node (BASELINE configs[3]).

The reference has no batch axis and no GPU; its concurrency model is *threads over independent calls* with
the GIL released (src/soxr_ext.cpp:222,297; tests/gil_bench.py:22-56).  The MI355X form of that model:

* `resample_batch(clips, in_rate, out_rate, quality, devices=None)` — ONE process, one host thread + one HIP
  stream per visible device; clips (any lengths: a corpus is ragged) are dealt to the devices in contiguous
  blocks (`shard`), each device resamples its block with ONE launch (ragged job table, include/hipsoxr.h
  `hipsoxr_job_t::clip_table`), results come back in the caller's order.  No data-path collective: clips are
  independent.
* one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI): `shard(n, world, rank)` names the
  rank's clips and `broadcast_bank(plan, ...)` is the path's only collective — the shared filter bank from rank 0
  at plan time, through torch's communicator or, with no torch in the path, through `hipsoxr_plan_broadcast`
  (C ABI, raw `ncclComm_t`).

bench.py and the multi-process tests import these from here; nothing below touches `oracle/`.
"""
import ctypes as _C
import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as _n
from . import device as _dev


def shard(n_units, world, rank):
    """Contiguous block partition of n_units independent clips over `world` ranks / devices: sizes differ by at
    most one, blocks are disjoint and cover [0, n_units)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    base, rem = divmod(int(n_units), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_bank(plan, group=None, root=0, comm=None, rank=None, stream=None, device=None):
    """Every rank ends up holding `root`'s float64 filter bank (the one collective of the path; `north_star`:
    "RCCL broadcast of the shared filter bank over xGMI").

    comm=None  : through torch.distributed (`group` or the default group).  Backend nccl (= RCCL): the bank
                 travels as a device tensor on `device` (default: the current HIP device); any other backend
                 (gloo: CPU tests) as a host tensor.  World size 1: nothing to do.
    comm=<int> : a raw ncclComm_t — `hipsoxr_plan_broadcast` (C ABI) does it on `stream`, no torch involved;
                 `rank` is this process's rank in that communicator."""
    if comm is not None:
        if rank is None:
            raise ValueError("broadcast_bank(comm=...) needs this process's rank in the communicator")
        _n.check(_n.lib.hipsoxr_plan_broadcast(plan.handle, _C.c_void_p(comm), int(root), int(rank),
                                               _C.c_void_p(stream) if stream else None))
        return
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    me = dist.get_rank(group)
    on_device = dist.get_backend(group) == "nccl"
    where = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_device \
        else torch.device("cpu")
    shape = (plan.phases, plan.taps, 4) if plan.phases else (plan.L, plan.taps)
    bank = torch.from_numpy(plan.bank()).to(where) if me == root else torch.empty(shape, dtype=torch.float64, device=where)
    dist.broadcast(bank, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
    if me != root:
        plan.set_bank(bank.cpu().numpy())


def bank_digest(plan):
    """Short SHA-256 of the bank a plan holds (equal on every rank after `broadcast_bank`)."""
    return hashlib.sha256(plan.bank().tobytes()).hexdigest()[:16]


def rank_info(plan, device=None, group=None):
    """What a multi-rank job really ran on: per rank the device index and name and the digest of the bank it holds.
    Collective (all ranks call it); every rank gets the list."""
    import torch
    import torch.distributed as dist
    dev_i = device.index if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else None)
    mine = {"rank": dist.get_rank(group) if dist.is_initialized() else 0, "device": dev_i,
            "name": torch.cuda.get_device_name(dev_i) if dev_i is not None else "cpu", "bank_sha256": bank_digest(plan)}
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {"ranks_seen": 1, "backend": None, "ranks": [mine], "devices_visible": visible, "banks_identical": True}
    got = [None] * dist.get_world_size(group)
    dist.all_gather_object(got, mine, group=group)
    backend = dist.get_backend(group)
    return {"ranks_seen": len(got), "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "ranks": got,
            "devices_visible": visible, "banks_identical": len({g["bank_sha256"] for g in got}) == 1}


class RaggedJob:
    """A batch of clips of unequal length on ONE device as one job: the clips packed end to end in a single buffer
    `[total_frames, channels]`, the per-clip table of `hipsoxr_job_t::clip_table` built once (host and device copy),
    `launch()` = one C call.  `outputs()` returns per-clip views of the packed result."""

    def __init__(self, plan, clips, kernel=_n.KERNEL_AUTO, stream=None, dither=None):
        """dither: TPDF dither on int16 output (None = on for int16, as `soxr_amd.resample` and libsoxr do; keyed by
        channel and output index within each clip, so a clip's result does not depend on its neighbours)."""
        import torch
        if not clips:
            raise ValueError("no clips")
        device = clips[0].device
        ch = 1 if clips[0].ndim == 1 else clips[0].shape[1]
        for c in clips:
            if c.device != device or c.dtype != clips[0].dtype or (1 if c.ndim == 1 else c.shape[1]) != ch or c.ndim not in (1, 2):
                raise ValueError("clips of one job share device, dtype and channel count; each is [frames] or [frames, channels]")
        n_in = [int(c.shape[0]) for c in clips]
        n_out = [plan.out_len(n) for n in n_in]
        self.x = torch.cat([c.reshape(-1, ch) for c in clips]).contiguous()  # packed [total_frames, channels]
        self.y = torch.empty((sum(n_out), ch), dtype=self.x.dtype, device=device)
        in_off = np.concatenate([[0], np.cumsum(n_in)[:-1]]) * ch
        out_off = np.concatenate([[0], np.cumsum(n_out)[:-1]]) * ch
        self._table = np.ascontiguousarray(np.stack([in_off, n_in, out_off, n_out], axis=1), dtype=np.int64)
        self._table_dev = torch.from_numpy(self._table).to(device)
        self.n_in, self.n_out, self._mono = n_in, n_out, [c.ndim == 1 for c in clips]
        j = _n.Job()
        j.in_, j.out, j.elem, j.kernel = self.x.data_ptr(), self.y.data_ptr(), _dev._torch_elem(self.x.dtype), kernel
        j.n_clips, j.n_channels = len(clips), ch
        j.in_clip_stride = j.out_clip_stride = 0
        j.in_frame_stride, j.in_chan_stride = ch, 1
        j.out_frame_stride, j.out_chan_stride = ch, 1
        j.in_abs0, j.in_frames, j.out_k0, j.out_frames = 0, max(n_in), 0, max(n_out)
        j.clip_table, j.clip_table_dev = self._table.ctypes.data, self._table_dev.data_ptr()
        j.dither = int(self.x.dtype == torch.int16 if dither is None else bool(dither))
        j.dither_seed = 0
        self._job, self._ref, self._plan = j, _C.byref(j), plan
        self._stream = stream if stream is not None else torch.cuda.current_stream(device).cuda_stream
        self._any = max(n_out) > 0

    def launch(self):
        if self._any:
            _n.check(_n.lib.hipsoxr_run_device(self._plan.handle, self._ref, self._stream))

    def outputs(self):
        outs, pos = [], 0
        for n, mono in zip(self.n_out, self._mono):
            v = self.y[pos:pos + n]
            outs.append(v[:, 0] if mono else v)
            pos += n
        return outs


_PLANS = {}


def _plan_on(device_index, in_rate, out_rate, quality, bank=None):
    """One plan per (device, conversion): device tables live where they were built.  Every device installs device
    0's bank (the in-process form of the bank broadcast: one design, identical coefficients everywhere)."""
    key = (device_index, float(in_rate), float(out_rate), str(quality))
    p = _PLANS.get(key)
    if p is None:
        p = _dev.Plan(in_rate, out_rate, quality)
        if bank is not None:
            p.set_bank(bank)
        _PLANS[key] = p
    return p


def resample_batch(clips, in_rate, out_rate, quality="VHQ", devices=None, kernel=_n.KERNEL_AUTO):
    """Resample independent clips on the GPUs of this node from ONE process.

    clips    : sequence of arrays, each [frames] or [frames, channels] — numpy (host) or torch tensors (any
               device); lengths may differ; dtype float32 / float64 / int16 / int32, the same for all.
    devices  : HIP device indices to use (default: all visible).  Clips are dealt in contiguous blocks
               (`shard(len(clips), len(devices), i)`); each device runs its block as ONE ragged launch on its
               own stream, driven by its own host thread (ctypes releases the GIL during every library call —
               the reference's threading model, tests/gil_bench.py:22-56).
    kernel   : engine selector for the device jobs (AUTO: the frequency-domain engine for large float jobs,
               1e-6-class; KERNEL_EXACT: the canonical-order engine, bit-identical to `soxr_amd.resample`).
    Returns a list of arrays of the same kind (numpy in -> numpy out; tensor in -> tensor on the device that
    computed it), in the order given."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device available (soxr_amd has no CPU fallback)")
    clips = list(clips)
    if not clips:
        return []
    if devices is None:
        devices = list(range(torch.cuda.device_count()))
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("no devices")
    bank0 = _plan_on(devices[0], in_rate, out_rate, quality).bank() if len(devices) > 1 else None
    results = [None] * len(clips)

    def work(i):
        lo, hi = shard(len(clips), len(devices), i)
        if lo == hi:
            return
        d = devices[i]
        with torch.cuda.device(d):
            plan = _plan_on(d, in_rate, out_rate, quality, bank0 if i else None)
            stream = torch.cuda.Stream(device=d)
            with torch.cuda.stream(stream):
                mine, was_numpy = [], []
                for c in clips[lo:hi]:
                    was_numpy.append(isinstance(c, np.ndarray))
                    t = torch.from_numpy(np.ascontiguousarray(c)) if isinstance(c, np.ndarray) else c
                    mine.append(t.to(torch.device("cuda", d), non_blocking=True))
                job = RaggedJob(plan, mine, kernel=kernel, stream=stream.cuda_stream)
                job.launch()
                outs = job.outputs()
                outs = [o.cpu().numpy() if w else o for o, w in zip(outs, was_numpy)]
            stream.synchronize()
        results[lo:hi] = outs

    if len(devices) == 1:
        work(0)
    else:
        with ThreadPoolExecutor(len(devices)) as ex:
            list(ex.map(work, range(len(devices))))
    return results
