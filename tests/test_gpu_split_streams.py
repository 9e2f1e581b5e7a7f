"""GPU: split-layout (one plane per channel) STREAMS fed small chunks — libsoxr's SOXR_SPLIT io spec, which the reference
binding drives per channel (src/soxr_ext.cpp:277-359).  A split stream whose first chunk is small runs interleaved inside
behind an adapter at the call boundary (csrc/engine.cpp hipsoxr_stream_process), so it gets the pinned host ring, the
resident kernel and deferred output like an interleaved stream.  Bars: frames bit-identical to the interleaved stream's,
call for call; with HIPSOXR_RESIDENT the per-call time within 25 % (+2 us) of the interleaved resident stream's."""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ptrs(addrs):
    return (C.c_void_p * len(addrs))(*addrs)


def _feed(n, x, in_rate, out_rate, elem, split, flags, chunk, recipe):
    """x: [frames, ch] C-contiguous (interleaved) — planes are taken from its transpose for the split stream.
    Returns (list of per-call frame counts, concatenated [frames, ch] result, seconds per call)."""
    frames, ch = x.shape
    planes = np.ascontiguousarray(x.T)
    item = x.itemsize
    h = C.c_void_p()
    n.check(n.lib.hipsoxr_stream_create(float(in_rate), float(out_rate), ch, elem | (4 if split else 0), recipe, flags, C.byref(h)))
    try:
        cap = int(chunk * out_rate / in_rate) + 64
        olen_total = int(frames * out_rate / in_rate) + 64
        done = C.c_size_t(0)
        counts, pos = [], 0
        if split:
            buf = np.zeros((ch, olen_total), x.dtype)
        else:
            buf = np.zeros((olen_total, ch), x.dtype)
        t0 = time.perf_counter()
        calls = 0
        for a in list(range(0, frames, chunk)) + [None]:
            while True:
                m = 0 if a is None else min(chunk, frames - a)
                if split:
                    ins = None if a is None else _ptrs([planes.ctypes.data + c * planes.strides[0] + a * item for c in range(ch)])
                    outs = _ptrs([buf.ctypes.data + c * buf.strides[0] + pos * item for c in range(ch)])
                else:
                    ins = None if a is None else x.ctypes.data + a * ch * item
                    outs = buf.ctypes.data + pos * ch * item
                n.check(n.lib.hipsoxr_stream_process(h, ins, m, outs, cap, C.byref(done)))
                counts.append(done.value); pos += done.value; calls += 1
                if a is not None or done.value == 0:
                    break
        dt = (time.perf_counter() - t0) / calls
        return counts, (buf.T[:pos] if split else buf[:pos]).copy(), dt
    finally:
        n.lib.hipsoxr_stream_delete(h)


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.float64])
@pytest.mark.parametrize("ch", [1, 2, 5])
def test_split_small_chunk_stream_equals_interleaved(dtype, ch):
    from soxr_amd import _native as n
    elem = {np.float32: n.F32, np.float64: n.F64, np.int16: n.I16}[dtype]
    rng = np.random.default_rng(5 + ch)
    x = rng.standard_normal((44100, ch)) * 0.25
    x = (x * 20000).astype(dtype) if dtype == np.int16 else x.astype(dtype)
    for flags in (0, n.RESIDENT, n.DEFER):
        ci, yi, _ = _feed(n, x, 44100, 16000, elem, False, flags, 441, 6)
        cs, ys, _ = _feed(n, x, 44100, 16000, elem, True, flags, 441, 6)
        assert ci == cs and np.array_equal(yi, ys), (dtype, ch, flags)


def test_split_stream_that_starts_large_keeps_its_planes_and_a_small_start_survives_large_chunks():
    from soxr_amd import _native as n
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((300000, 3)) * 0.25).astype(np.float32)
    elem = n.F32
    want = _feed(n, x, 48000, 44100, elem, False, 0, 100000, 6)[1]
    assert np.array_equal(_feed(n, x, 48000, 44100, elem, True, 0, 100000, 6)[1], want)       # planar device ring throughout
    # small first chunk (adapter on), then the rest in one large chunk
    planes = np.ascontiguousarray(x.T)
    h = C.c_void_p()
    n.check(n.lib.hipsoxr_stream_create(48000., 44100., 3, elem | 4, 6, 0, C.byref(h)))
    try:
        out = np.zeros((3, 300000), np.float32)
        done, pos = C.c_size_t(0), 0
        for a, m in ((0, 480), (480, 300000 - 480), (None, 0), (None, 0)):
            ins = None if a is None else _ptrs([planes.ctypes.data + c * planes.strides[0] + a * 4 for c in range(3)])
            outs = _ptrs([out.ctypes.data + c * out.strides[0] + pos * 4 for c in range(3)])
            n.check(n.lib.hipsoxr_stream_process(h, ins, m, outs, 300000 - pos, C.byref(done)))
            pos += done.value
        assert pos == want.shape[0] and np.array_equal(out.T[:pos], want)
    finally:
        n.lib.hipsoxr_stream_delete(h)


def test_split_resident_stream_is_as_fast_as_the_interleaved_one():
    from soxr_amd import _native as n
    elem = n.I16
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((44100 * 4, 2)) * 5000).astype(np.int16)
    best = {}
    for split in (False, True):
        best[split] = min(_feed(n, x, 44100, 16000, elem, split, n.RESIDENT, 441, 6)[2] for _ in range(3))
    assert best[True] < 1.25 * best[False] + 2e-6, best
