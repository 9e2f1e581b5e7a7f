"""Batches of independent clips over the GPUs of one node (BASELINE configs[3]).

The reference has no batch axis and no GPU; its concurrency model is *threads over independent calls* with
the GIL released (src/soxr_ext.cpp:222,297; tests/gil_bench.py:22-56).  The MI355X form of that model:

* `resample_batch(clips, in_rate, out_rate, quality, devices=None)` — ONE process, one host thread per visible
  device; clips (any lengths: a corpus is ragged) are dealt to the devices by total frames (`shard_by_frames`), each
  device resamples its share as ragged launches (job table, include/hipsoxr.h `hipsoxr_job_t::clip_table`), results
  come back in the caller's order.  Device tensors are resampled where they lie (the table addresses them in
  place); host arrays go through a pinned staging ring with the H2D copy of block k+1, the launch of block k and
  the D2H copy of block k-1 overlapped on three streams.  No data-path collective: clips are independent.
* one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI): `shard(n, world, rank)` names the
  rank's clips and `broadcast_bank(plan, ...)` is the path's only collective — the shared filter bank from rank 0
  at plan time, through torch's communicator or, with no torch in the path, through `hipsoxr_plan_broadcast`
  (C ABI, raw `ncclComm_t`).

bench.py and the multi-process tests import these from here; nothing below touches `oracle/`.
"""
import ctypes as _C
import hashlib
import threading
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


def shard_by_frames(lengths, world):
    """Partition clips of unequal length over `world` devices so that the devices' TOTAL frames are as even as greedy
    longest-first assignment makes them (a ragged corpus dealt by count loads devices by luck).  Returns `world` lists
    of clip indices (each ascending); every index appears exactly once.  Deterministic."""
    if world < 1:
        raise ValueError("need world >= 1")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads, parts = [0] * world, [[] for _ in range(world)]
    for i in order:
        w = min(range(world), key=lambda k: (loads[k], k))
        parts[w].append(i)
        loads[w] += int(lengths[i])
    return [sorted(p) for p in parts]


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
    """A batch of clips of unequal length on ONE device as one job.  Clips that are already device tensors are
    resampled WHERE THEY LIE: the per-clip table of `hipsoxr_job_t::clip_table` addresses each one relative to the
    lowest clip (no packed copy of the input); results go to one packed buffer `[total_out_frames, channels]`.
    `launch()` = one C call.  `outputs()` returns per-clip views of the packed result.

    Stream order: the job's tensors are produced on the caller's current stream; when it launches on another stream
    that stream first waits for the current one (and the result records its use there)."""

    def __init__(self, plan, clips, kernel=_n.KERNEL_AUTO, stream=None, dither=None):
        """dither: TPDF dither on int16 output (None = on for int16, as `soxr_amd.resample` and libsoxr do; keyed by
        channel and output index within each clip, so a clip's result does not depend on its neighbours).
        stream: a torch.cuda.Stream (ordering handled here) or a raw hipStream_t handle (the caller orders it)."""
        import torch
        if not clips:
            raise ValueError("no clips")
        device = clips[0].device
        ch = 1 if clips[0].ndim == 1 else clips[0].shape[1]
        for c in clips:
            if c.device != device or c.dtype != clips[0].dtype or (1 if c.ndim == 1 else c.shape[1]) != ch or c.ndim not in (1, 2):
                raise ValueError("clips of one job share device, dtype and channel count; each is [frames] or [frames, channels]")
        es = clips[0].element_size()
        keep = [c if c.is_contiguous() else c.contiguous() for c in clips]  # (a strided clip is the one case that is copied)
        n_in = [int(c.shape[0]) for c in keep]
        n_out = [plan.out_len(n) for n in n_in]
        ptrs = [c.data_ptr() for c in keep if c.shape[0] > 0]   # (an empty tensor has no storage address)
        self.y = torch.empty((sum(n_out), ch), dtype=keep[0].dtype, device=device)
        base = min(ptrs) if ptrs else self.y.data_ptr()
        in_off = np.array([(c.data_ptr() - base) // es if c.shape[0] > 0 else 0 for c in keep], dtype=np.int64)
        out_off = np.concatenate([[0], np.cumsum(n_out)[:-1]]).astype(np.int64) * ch
        self._table = np.ascontiguousarray(np.stack([in_off, n_in, out_off, n_out], axis=1), dtype=np.int64)
        self._table_dev = torch.from_numpy(self._table).to(device)
        self.n_in, self.n_out, self._mono, self._keep = n_in, n_out, [c.ndim == 1 for c in clips], keep
        j = _n.Job()
        j.in_, j.out, j.elem, j.kernel = base, self.y.data_ptr(), _dev._torch_elem(keep[0].dtype), kernel
        j.n_clips, j.n_channels = len(clips), ch
        j.in_clip_stride = j.out_clip_stride = 0
        j.in_frame_stride, j.in_chan_stride = ch, 1
        j.out_frame_stride, j.out_chan_stride = ch, 1
        j.in_abs0, j.in_frames, j.out_k0, j.out_frames = 0, max(n_in), 0, max(n_out)
        j.clip_table, j.clip_table_dev = self._table.ctypes.data, self._table_dev.data_ptr()
        j.dither = int(keep[0].dtype == torch.int16 if dither is None else bool(dither))
        j.dither_seed = 0
        self._job, self._ref, self._plan = j, _C.byref(j), plan
        cur = torch.cuda.current_stream(device)
        if isinstance(stream, torch.cuda.Stream):
            stream.wait_stream(cur)           # y, the table and the clips themselves were produced on the current stream
            self.y.record_stream(stream)
            for c in keep:
                c.record_stream(stream)
            self._stream = stream.cuda_stream
        else:
            self._stream = stream if stream is not None else cur.cuda_stream
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


# Plans per (device, conversion), least recently used first: device tables live where they were built, and a long-lived
# service that meets many rate pairs must not keep all of them for ever.
_PLANS = {}
_PLANS_MAX = 32
_CACHE_LOCK = threading.Lock()   # guards _PLANS and _PIPES: `resample_batch` is called from many threads (the module's model)


def _plan_on(device_index, in_rate, out_rate, quality, bank=None):
    """One plan per (device, conversion).  Every device installs device 0's bank (the in-process form of the bank
    broadcast: one design, identical coefficients everywhere).  At most `_PLANS_MAX` plans are kept (LRU);
    `clear_plans()` drops them all."""
    key = (device_index, float(in_rate), float(out_rate), str(quality))
    with _CACHE_LOCK:
        p = _PLANS.pop(key, None)
        if p is None:
            p = _dev.Plan(in_rate, out_rate, quality)
            if bank is not None:
                p.set_bank(bank)
        _PLANS[key] = p                      # (re-inserted: most recently used last)
        while len(_PLANS) > _PLANS_MAX:
            _PLANS.pop(next(iter(_PLANS)))
    return p


def clear_plans():
    """Drop the cached plans (and with them their device tables) and the idle staging pipes."""
    with _CACHE_LOCK:
        _PLANS.clear()
        _PIPES.clear()


class _PipeRun:
    """Bookkeeping of ONE `_HostPipe.run`: the block list, the events between its three host threads and the first
    error.  `fail()` wakes everything that can wait, so that an error anywhere — a HIP call in the issuing loop, an
    allocation in a copy thread — ends the run as an exception in the caller instead of a hang."""

    def __init__(self, n_in, n_out, ch, es, block_bytes, slots):
        self.n_in, self.n_out, self.ch, self.slots = n_in, n_out, ch, slots
        self.blocks, cur, cur_b = [], [], 0          # blocks hold whole clips, about block_bytes of input each
        for i, n in enumerate(n_in):
            if cur and cur_b + n * ch * es > block_bytes:
                self.blocks.append(cur); cur, cur_b = [], 0
            cur.append(i); cur_b += n * ch * es
        if cur:
            self.blocks.append(cur)
        nb = len(self.blocks)
        self.tot_in = [sum(n_in[i] for i in b) * ch for b in self.blocks]
        self.tot_out = [sum(n_out[i] for i in b) * ch for b in self.blocks]
        self.ev_free_in = [None] * slots    # h2d of the block that used the slot before has read the pinned input
        self.ev_out = [None] * nb           # d2h(k) done
        self.ev_run = [None] * nb           # run(k) done
        self.free_out = [threading.Semaphore(1) for _ in range(slots)]   # the output slot has been unstaged
        self.own_out = [None] * nb          # pinned_results: the block's own pinned result buffer
        self.staged = [threading.Event() for _ in range(nb)]
        self.issued = [threading.Event() for _ in range(nb)]
        self.err = []

    def fail(self, e):
        self.err.append(e)
        for ev in self.staged + self.issued:
            ev.set()
        for sem in self.free_out:
            sem.release()


class _HostPipe:
    """Host clips of one device through a pinned staging ring: three streams, SLOTS block slots.
         stage(k)   CPU copies block k's clips into the pinned input slot      (host thread A)
         h2d(k)     pinned slot -> packed device buffer                        (stream `sin`)
         run(k)     ONE ragged launch over the block                           (stream `sc`, behind h2d(k))
         d2h(k)     packed device result -> pinned output slot                 (stream `sout`, behind run(k))
         unstage(k) CPU copies the slot into the caller's result arrays        (host thread B, behind d2h(k))
    so that the H2D copy of block k+1, the launch of block k and the D2H copy of block k-1 are in flight together and
    the two CPU copies run beside them.  Blocks hold whole clips, about `block_bytes` of input each.
    With `pinned_results` d2h(k) lands in a pinned buffer of the block's own (from torch's caching host allocator) and the
    result arrays ARE views of it: no unstage copy — half of the CPU work of the path, which is what bounds it.
    One run at a time per pipe (`resample_batch` checks pipes out of a pool: concurrent callers get pipes of their own)."""
    SLOTS = 3

    def __init__(self, plan, device, dtype, ch, kernel, block_bytes, pinned_results=False):
        import os
        import torch
        self.torch, self.plan, self.device, self.kernel, self.ch = torch, plan, device, kernel, ch
        self.pinned_results = pinned_results
        self.tdtype = dtype
        self.sin, self.sc, self.sout = (torch.cuda.Stream(device=device) for _ in range(3))
        self.block_bytes = block_bytes
        self.copy_threads = int(os.environ.get("SOXR_AMD_COPY_THREADS", 0)) or max(2, min(8, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4) // 2))
        self.cap_in = self.cap_out = 0
        self.pin_in = self.pin_out = self.dev_in = self.dev_out = None

    def _ensure(self, n_in_el, n_out_el):
        torch = self.torch
        n_in_el, n_out_el = max(1, n_in_el), max(1, n_out_el)   # (a share of zero-length clips still has slots to name)
        if n_in_el > self.cap_in:
            self.cap_in = n_in_el
            self.pin_in = [torch.empty(n_in_el, dtype=self.tdtype, pin_memory=True) for _ in range(self.SLOTS)]
            self.dev_in = [torch.empty(n_in_el, dtype=self.tdtype, device=self.device) for _ in range(self.SLOTS)]
        if n_out_el > self.cap_out:
            self.cap_out = n_out_el
            self.pin_out = None
            self.dev_out = [torch.empty(n_out_el, dtype=self.tdtype, device=self.device) for _ in range(self.SLOTS)]
        if not self.pinned_results and (self.pin_out is None or self.pin_out[0].numel() < self.cap_out):
            self.pin_out = [torch.empty(self.cap_out, dtype=self.tdtype, pin_memory=True) for _ in range(self.SLOTS)]

    def _stager(self, st, clips, pool):
        """Host thread A: block k's clips into pinned input slot k mod SLOTS, spread over the copy threads."""
        try:
            for k, b in enumerate(st.blocks):
                s = k % self.SLOTS
                if k >= self.SLOTS:               # the pinned slot's previous block has been copied to the device
                    st.issued[k - self.SLOTS].wait()
                    if st.err:
                        return
                    st.ev_free_in[s].synchronize()
                view, pos, runs = self.pin_in[s].numpy(), 0, [[] for _ in range(self.copy_threads)]
                share, t = st.tot_in[k] / self.copy_threads, 0
                for i in b:                       # one task per copy thread: a run of clips of about equal bytes
                    n = st.n_in[i] * st.ch
                    if pos >= (t + 1) * share and t + 1 < self.copy_threads:
                        t += 1
                    runs[t].append((pos, n, i))
                    pos += n

                def put(run):
                    for pos, n, i in run:
                        np.copyto(view[pos:pos + n], clips[i].reshape(-1), casting="no")

                for f in [pool.submit(put, r) for r in runs if r]:
                    f.result()
                st.staged[k].set()
        except Exception as e:
            st.fail(e)

    def _unstager(self, st, clips, results, idx, pool):
        """Host thread B: block k's results out of its pinned buffer (views of it with `pinned_results`)."""
        ch = st.ch
        try:
            for k, b in enumerate(st.blocks):
                st.issued[k].wait()
                if st.err:
                    return
                st.ev_out[k].synchronize()
                s = k % self.SLOTS
                shape = lambda i: (st.n_out[i],) if clips[i].ndim == 1 else (st.n_out[i], ch)   # noqa: E731
                if st.own_out[k] is not None:     # the results are views of the block's pinned buffer (which they keep alive)
                    view, pos = st.own_out[k].numpy(), 0
                    for i in b:
                        n = st.n_out[i] * ch
                        results[idx[i]] = view[pos:pos + n].reshape(shape(i))
                        pos += n
                    st.own_out[k] = None
                    continue
                view, pos, jobs = self.pin_out[s].numpy(), 0, []

                def take(i, pos, n):   # (allocation included: the first touch of a fresh result array is most of its cost)
                    out = np.empty(shape(i), dtype=clips[i].dtype)
                    np.copyto(out.reshape(-1), view[pos:pos + n], casting="no")
                    results[idx[i]] = out

                for i in b:
                    n = st.n_out[i] * ch
                    jobs.append(pool.submit(take, i, pos, n))
                    pos += n
                for f in jobs:
                    f.result()
                st.free_out[s].release()
        except Exception as e:
            st.fail(e)

    def _issue(self, st, k):
        """Block k onto the three streams: h2d on `sin`, the ragged launch on `sc`, d2h on `sout`."""
        torch, ch, s, b = self.torch, st.ch, k % self.SLOTS, st.blocks[k]
        tot_in, tot_out = st.tot_in[k], st.tot_out[k]
        with torch.cuda.stream(self.sin):
            if k >= self.SLOTS:
                self.sin.wait_event(st.ev_run[k - self.SLOTS])   # dev_in[s] was read by run(k - SLOTS)
            self.dev_in[s][:tot_in].copy_(self.pin_in[s][:tot_in], non_blocking=True)
            e_in = torch.cuda.Event(); e_in.record(self.sin)
        st.ev_free_in[s] = e_in
        # the block's table: clips end to end in the packed device buffers (device copy uploaded by the library)
        ni = np.array([st.n_in[i] for i in b], np.int64); no = np.array([st.n_out[i] for i in b], np.int64)
        table = np.ascontiguousarray(np.stack([np.concatenate([[0], np.cumsum(ni)[:-1]]) * ch, ni,
                                               np.concatenate([[0], np.cumsum(no)[:-1]]) * ch, no], axis=1), dtype=np.int64)
        j = _n.Job()
        j.in_, j.out = self.dev_in[s].data_ptr(), self.dev_out[s].data_ptr()
        j.elem, j.kernel = _dev._torch_elem(self.tdtype), self.kernel
        j.n_clips, j.n_channels = len(b), ch
        j.in_frame_stride, j.in_chan_stride, j.out_frame_stride, j.out_chan_stride = ch, 1, ch, 1
        j.in_frames, j.out_frames = int(ni.max()), int(no.max())
        j.clip_table, j.clip_table_dev = table.ctypes.data, None
        j.dither = int(self.tdtype == torch.int16)
        if self.pinned_results:
            st.own_out[k] = torch.empty(max(tot_out, 1), dtype=self.tdtype, pin_memory=True)
        else:
            st.free_out[s].acquire()              # the result slot of block k - SLOTS has been copied out
            if st.err:
                return
        self.sc.wait_event(e_in)
        if k >= self.SLOTS:
            self.sc.wait_event(st.ev_out[k - self.SLOTS])   # ... and its device buffer read by d2h(k - SLOTS)
        if j.out_frames > 0:
            _n.check(_n.lib.hipsoxr_run_device(self.plan.handle, _C.byref(j), self.sc.cuda_stream))
        e_c = torch.cuda.Event(); e_c.record(self.sc)
        st.ev_run[k] = e_c
        with torch.cuda.stream(self.sout):
            self.sout.wait_event(e_c)
            (st.own_out[k] if self.pinned_results else self.pin_out[s])[:tot_out].copy_(self.dev_out[s][:tot_out], non_blocking=True)
            e_o = torch.cuda.Event(); e_o.record(self.sout)
        st.ev_out[k] = e_o

    def run(self, clips, results, idx):
        """clips: numpy arrays of this device (all the same dtype / channel count); results[idx[i]] = resampled clips[i].
        Any error — in the issuing loop or in either copy thread — is raised here after all three have stopped."""
        n_in = [int(c.shape[0]) for c in clips]
        st = _PipeRun(n_in, [self.plan.out_len(n) for n in n_in], self.ch, clips[0].dtype.itemsize, self.block_bytes, self.SLOTS)
        self._ensure(max(st.tot_in), max(st.tot_out))
        # the two CPU copies of a block are spread over a few threads (numpy releases the GIL inside a large copy; one
        # thread moves ~10-25 GB/s, the link 63 GB/s each way)
        pool = ThreadPoolExecutor(self.copy_threads)
        ta = threading.Thread(target=self._stager, args=(st, clips, pool))
        tb = threading.Thread(target=self._unstager, args=(st, clips, results, idx, pool))
        ta.start(); tb.start()
        try:
            for k in range(len(st.blocks)):
                st.staged[k].wait()
                if st.err:
                    break
                self._issue(st, k)
                if st.err:
                    break
                st.issued[k].set()
        except Exception as e:
            st.fail(e)
        finally:
            for ev in st.issued:
                ev.set()
            ta.join(); tb.join()
            pool.shutdown()
        if st.err:
            for s_ in (self.sin, self.sc, self.sout):   # nothing of the failed run may still be in flight when the slots are reused
                try:
                    s_.synchronize()
                except Exception:  # noqa: BLE001 — (the error being raised is the first one)
                    pass
            raise st.err[0]


_PIPES = {}     # key -> idle pipes (a run checks one out for its duration: concurrent callers never share slots)


def _checkout_pipe(key, make):
    with _CACHE_LOCK:
        idle = _PIPES.get(key)
        if idle:
            return idle.pop()
    return make()


def _return_pipe(key, pipe, keep=2):
    with _CACHE_LOCK:
        idle = _PIPES.setdefault(key, [])
        if len(idle) < keep:           # (more than that were made for a burst of concurrent callers: let them go)
            idle.append(pipe)


PINNED_RESULTS_MAX = 8 << 30     # host results up to this many bytes per call are returned in pinned memory by default


def resample_batch(clips, in_rate, out_rate, quality="VHQ", devices=None, kernel=_n.KERNEL_AUTO, block_bytes=64 << 20,
                   pinned_results=None):
    """Resample independent clips on the GPUs of this node from ONE process.

    clips    : sequence of arrays, each [frames] or [frames, channels] — numpy (host) or torch tensors (any
               device); lengths may differ; dtype float32 / float64 / int16 / int32, the same for all.
    devices  : HIP device indices to use (default: all visible).  Clips are dealt by total frames
               (`shard_by_frames`); each device is driven by its own host thread (ctypes releases the GIL during
               every library call — the reference's threading model, tests/gil_bench.py:22-56).
               Device tensors: one ragged launch per device over the clips WHERE THEY LIE (a clip on another device
               is copied over first), on a side stream ordered behind the caller's current streams; the caller's
               current stream on the computing device waits for the results.
               Host arrays: a pinned staging ring per device, blocks of about `block_bytes` of input, with the H2D
               copy of block k+1, the launch of block k and the D2H copy of block k-1 in flight together.
    pinned_results : host results as views of page-locked buffers the D2H copies land in (one buffer per block, kept
               alive by its arrays, recycled by torch's caching host allocator once they are all dropped) instead of
               fresh pageable arrays filled by one more CPU copy.  None: yes while the call's results stay under
               PINNED_RESULTS_MAX bytes.  The arrays are ordinary writable numpy arrays either way.
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
    if pinned_results is None:
        r = float(out_rate) / float(in_rate)
        pinned_results = sum(c.nbytes for c in clips if isinstance(c, np.ndarray)) * r <= PINNED_RESULTS_MAX
    parts = shard_by_frames([int(c.shape[0]) for c in clips], len(devices))

    # torch's current stream is THREAD-local: the workers below would see their own default streams, not the caller's.
    # The streams the caller's tensors were produced on (and on which the results must be ordered) are looked up HERE.
    involved = set(devices) | {c.device.index for c in clips if not isinstance(c, np.ndarray) and c.is_cuda}
    caller = {d: torch.cuda.current_stream(torch.device("cuda", d)) for d in involved}

    def work(i):
        mine = parts[i]
        if not mine:
            return
        d = devices[i]
        dev_t = torch.device("cuda", d)
        with torch.cuda.device(d):
            plan = _plan_on(d, in_rate, out_rate, quality, bank0 if i else None)
            host = [k for k in mine if isinstance(clips[k], np.ndarray)]
            dev = [k for k in mine if not isinstance(clips[k], np.ndarray)]
            if host and sum(int(clips[k].shape[0]) for k in host) == 0:   # nothing to move: empty results of the right shape
                for k in host:
                    results[k] = np.empty((0,) + clips[k].shape[1:], clips[k].dtype)
                host = []
            if host:
                c0 = clips[host[0]]
                ch = 1 if c0.ndim == 1 else c0.shape[1]
                tdt = torch.from_numpy(np.empty(0, c0.dtype)).dtype
                key = (d, tdt, ch, int(kernel), int(block_bytes), bool(pinned_results))
                pipe = _checkout_pipe(key, lambda: _HostPipe(plan, dev_t, tdt, ch, kernel, block_bytes, bool(pinned_results)))
                pipe.plan = plan
                try:
                    pipe.run([np.ascontiguousarray(clips[k]) for k in host], results, host)
                finally:
                    _return_pipe(key, pipe)
            if dev:
                cur = caller[d]
                side = torch.cuda.Stream(device=dev_t)
                tens = []
                for k in dev:
                    c = clips[k]
                    if c.is_cuda:   # whatever produced the clip on its own device's (caller's) current stream comes first
                        side.wait_stream(caller[c.device.index])
                    with torch.cuda.stream(side):
                        tens.append(c if c.device == dev_t else c.to(dev_t, non_blocking=True))
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    job = RaggedJob(plan, tens, kernel=kernel, stream=side.cuda_stream)
                    job.launch()
                    outs = job.outputs()
                for t in tens + [job.y, job._table_dev]:
                    t.record_stream(side)
                cur.wait_stream(side)   # the caller's stream sees finished results (no host synchronisation needed)
                for k, o in zip(dev, outs):
                    results[k] = o

    if len(devices) == 1:
        work(0)
    else:
        with ThreadPoolExecutor(len(devices)) as ex:
            list(ex.map(work, range(len(devices))))
    return results
