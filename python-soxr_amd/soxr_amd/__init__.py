"""soxr_amd — MI355X-native drop-in for python-soxr's resampling surface.

Mirrors the reference package `soxr` (src/soxr/__init__.py): `resample`, `ResampleStream`,
`_resample_oneshot`, the quality constants `QQ LQ MQ HQ VHQ`, `__version__` and
`__libsoxr_version__` — same argument meaning, same output shape/dtype/length, same exception
classes in the same order.  The arithmetic runs in hand-written HIP kernels behind the C ABI of
include/hipsoxr.h (libhipsoxr.so, loaded with ctypes); there is no CPU fallback.

    import soxr_amd as soxr
    y = soxr.resample(x, 48000, 44100, quality="VHQ")
"""
import ctypes as _C
import weakref as _weakref

import numpy as np

from . import _native as _n
from ._native import QQ, LQ, MQ, HQ, VHQ

__libsoxr_version__ = _n.version()  # reference: soxr_ext.libsoxr_version(), src/soxr/__init__.py:18
# one version number: the native library's (include/hipsoxr.h HIPSOXR_VERSION_STRING; setup.py gives the wheel the same)
__version__ = __libsoxr_version__.split("-", 1)[-1].split(" ", 1)[0]   # "hipsoxr-0.6.0 (gfx950)" -> "0.6.0"


def prefix():
    """Install prefix of the libsoxr-named ABI (lib/libsoxr.so, include/soxr.h, lib/pkgconfig/soxr.pc):
    what a libsoxr client's CMAKE_PREFIX_PATH should point at, e.g. the reference's own
    `-DUSE_SYSTEM_LIBSOXR=ON` build (CMakeLists.txt:29,83-93).  Present in an installed wheel; in a
    source checkout the header lives in include/ and the library beside this file."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "prefix")


# same limit as the reference (src/soxr/__init__.py:22)
_CH_LIMIT = 65536

# One device job covers a whole signal up to this many frames; longer inputs go through the
# streaming path in pieces (bounded device memory).  The reference's analogue is `div_len`
# (src/soxr_ext.cpp:239), which exists for a different reason (libsoxr slows down on long input).
_DIV_FRAMES = 1 << 24

_QUALITY_NAMES = {"qq": QQ, "lq": LQ, "mq": MQ, "hq": HQ, "vhq": VHQ}
_DTYPE_CODES = {np.dtype(np.float32): _n.F32, np.dtype(np.float64): _n.F64,
                np.dtype(np.int32): _n.I32, np.dtype(np.int16): _n.I16}
_SUPPORTED = "[float32, float64, int16, int32]"


def _quality_to_enum(q):
    """'HQ' / 'soxr_hq' / soxr.HQ -> recipe int; anything else -> ValueError
    (reference: src/soxr/__init__.py:38-45)."""
    if isinstance(q, str):
        key = q.lower()
        if key.startswith("soxr_"):
            key = key[5:]
        if key in _QUALITY_NAMES:
            return _QUALITY_NAMES[key]
    elif isinstance(q, (int, np.integer)) and not isinstance(q, bool):
        if int(q) in (QQ, LQ, MQ, HQ, VHQ):
            return int(q)
    raise ValueError("Quality must be one of [QQ, LQ, MQ, HQ, VHQ]")


def _elem_code(dtype):
    try:
        return _DTYPE_CODES[np.dtype(dtype)]
    except (KeyError, TypeError):
        raise TypeError(f"Data type must be one of {_SUPPORTED}, not {dtype}")


def _check_rates(in_rate, out_rate):
    if in_rate <= 0 or out_rate <= 0:
        raise ValueError("Sample rate should be over 0")


def _check_channels(n):
    if n < 1 or n > _CH_LIMIT:
        raise ValueError(f"Channel num({n}) out of limit. Should be in [1, {_CH_LIMIT}]")


def _ptr_array(ptrs):
    arr = (_C.c_void_p * len(ptrs))(*ptrs)
    return arr


class _CSoxrView:
    """What `ResampleStream._csoxr` is in the reference: the binding's handle object
    (nb::class_<CSoxr>, src/soxr_ext.cpp:408-423).  The reference's own tests reach through it
    (`rs._csoxr.engine()`, tests/conftest.py:7,10); here it is a view of the stream, same attribute and
    method names, arguments in the binding's units (io_ratio, not two rates)."""

    def __init__(self, rs, in_rate, out_rate):
        self._rs = _weakref.proxy(rs)  # (no reference cycle: the stream's __del__ releases device state promptly)
        self.in_rate, self.out_rate = float(in_rate), float(out_rate)
        self.channels = rs._channels
        self.ntype = _elem_code(rs._type)  # SOXR_*_I numbering (src/soxr_ext.cpp:35-38)

    @property
    def ended(self):
        return self._rs._ended

    def _proc(self, dtype, x, last):
        if self._rs._type != np.dtype(dtype):  # (nanobind would refuse the ndarray<T> argument)
            raise TypeError(f"process_{dtype}: stream was created for {self._rs._type}")
        return self._rs._process(x, last)

    def process_float32(self, x, last=False):
        return self._proc("float32", x, last)

    def process_float64(self, x, last=False):
        return self._proc("float64", x, last)

    def process_int32(self, x, last=False):
        return self._proc("int32", x, last)

    def process_int16(self, x, last=False):
        return self._proc("int16", x, last)

    def num_clips(self):
        return self._rs.num_clips()

    def delay(self):
        return self._rs.delay()

    def engine(self):
        return self._rs.engine()

    def clear(self):
        self._rs.clear()

    def set_io_ratio(self, io_ratio, slew_len=0):  # (src/soxr_ext.cpp:200-204)
        if io_ratio <= 0:
            raise ValueError("Sample rate should be over 0")
        self._rs.set_io_ratio(float(io_ratio), 1.0, slew_len)


class ResampleStream:
    """Streaming resampler: state (pending input, counters) lives on the GPU between calls.

    Mirrors soxr.ResampleStream (src/soxr/__init__.py:61-179) and the CSoxr handle beneath it
    (src/soxr_ext.cpp:49-205).

    Parameters
    ----------
    in_rate, out_rate : float       sample rates (> 0)
    num_channels : int              1 .. 65536
    dtype : float32 | float64 | int16 | int32 (type or str)
    quality : 'QQ' | 'LQ' | 'MQ' | 'HQ' | 'VHQ' (or the soxr.* constants)
    vr : bool                       (experimental in the reference) variable-rate mode: in_rate/out_rate
                                    must be the LARGEST io ratio that will be used; see set_io_ratio
    deferred : bool                 (extension) deferred output: `resample_chunk` enqueues its copy and kernel
                                    and returns the frames the PREVIOUS call produced, so no GPU round trip
                                    sits inside the call (12-15 us instead of ~36 us for 10 ms chunks).  The
                                    concatenated output is identical; frames surface one call later — the
                                    reference's contract allows any per-call count (README.md:77-78) — and
                                    `delay()` counts them as pending.  Constant-rate streams only.
    resident : bool                 (extension) synchronous calls on small chunks (up to 2048 result frames) are
                                    served by a kernel that stays on the GPU between calls and is fed through a
                                    mailbox in pinned memory: no HIP call per chunk (10 ms chunks: ~12 us per
                                    call instead of ~31 us).  Output identical, frames surface in the same
                                    call.  The kernel leaves by itself 1 ms after the last call
                                    (HIPSOXR_RESIDENT_IDLE_US).  Interleaved streams (constant or variable rate), not with
                                    `deferred`.  resident="auto" (or environment HIPSOXR_AUTO_RESIDENT): the stream turns
                                    this path on by itself after 16 small back-to-back calls and off again when the
                                    run breaks.  Opt-in, because while a resident kernel spins every device-wide
                                    synchronisation in the process (torch.cuda.synchronize, hipFree) waits for it to
                                    leave.
    dither_seed : int               (extension) seed of the int16 TPDF dither.  libsoxr seeds randomly per
                                    handle; here dither is a deterministic function of (seed, channel,
                                    output index), default seed 0 — pass distinct seeds to decorrelate
                                    concurrent streams
    """

    def __init__(self, in_rate, out_rate, num_channels, dtype="float32", quality="HQ", vr=False, dither_seed=0,
                 deferred=False, resident=False):
        _check_rates(in_rate, out_rate)
        _check_channels(num_channels)
        self._type = np.dtype(dtype)
        elem = _elem_code(self._type)
        recipe = _quality_to_enum(quality)
        self._channels = int(num_channels)
        self._ratio = float(out_rate) / float(in_rate)
        self._h = _C.c_void_p()
        flags = ((_n.VR if vr else 0) | (_n.DEFER if deferred and not vr else 0)
                 | ((_n.AUTO_RESIDENT if resident == "auto" else _n.RESIDENT) if resident and not deferred else 0))
        _n.check(_n.lib.hipsoxr_stream_create(float(in_rate), float(out_rate), self._channels,
                                              elem, recipe, flags, _C.byref(self._h)))
        if dither_seed:
            _n.check(_n.lib.hipsoxr_stream_set_dither_seed(self._h, int(dither_seed) & 0xFFFFFFFF))
        self._ended = False
        self._csoxr = _CSoxrView(self, in_rate, out_rate)  # (src/soxr/__init__.py:99)
        self._done = _C.c_size_t(0)
        self._done_ref = _C.byref(self._done)
        # Output capacity of a call without asking the library for delay(): after a synchronous call the frames
        # still pending are those whose filter support reaches past the input, at most (taps/2 + 2) * out/in + 1.
        # (Variable-rate and deferred streams ask: their backlog is not bounded like that.)
        self._slack = None
        if not vr and not deferred:
            info = _n.PlanInfo()
            _n.check(_n.lib.hipsoxr_plan_info(_n.lib.hipsoxr_stream_plan(self._h), _C.byref(info)))
            self._slack = int((info.taps / 2 + 2) * self._ratio) + 4

    def __del__(self, _delete=_n.lib.hipsoxr_stream_delete):  # bound early: globals may be gone at exit
        h = getattr(self, "_h", None)
        if h:
            _delete(h)
            self._h = None

    # -- the counterpart of CSoxr::process (src/soxr_ext.cpp:129-188) ---------------------------
    def _process(self, x, last, mono=False):
        """x: 2-D [frame, channel], or 1-D when mono (the result then is 1-D too)."""
        if self._ended:
            raise RuntimeError("Input after last input")
        if (1 if mono else x.shape[1]) != self._channels:
            raise ValueError("Channel num mismatch")
        if not x.flags.c_contiguous:
            x = np.ascontiguousarray(x)
        frames = x.shape[0]
        if self._slack is not None:
            cap = int(frames * self._ratio) + self._slack
        else:
            cap = int(_n.lib.hipsoxr_stream_delay(self._h) + frames * self._ratio) + 2
        y = np.empty(cap if mono else (cap, self._channels), self._type)
        done = self._done
        # a zero-length chunk still drains (in != NULL, ilen == 0)
        in_ptr = x.ctypes.data if frames else y.ctypes.data
        err = _n.lib.hipsoxr_stream_process(self._h, in_ptr, frames, y.ctypes.data, cap, self._done_ref)
        if err:
            _n.check(err)
        pos = done.value
        if last:
            self._ended = True
            row = self._channels * self._type.itemsize
            while True:  # flush until the stream runs dry (src/soxr_ext.cpp:109-127)
                if pos >= y.shape[0]:
                    y = np.concatenate([y, np.empty_like(y)])
                _n.check(_n.lib.hipsoxr_stream_process(self._h, None, 0, y.ctypes.data + pos * row,
                                                       y.shape[0] - pos, self._done_ref))
                if done.value == 0:
                    break
                pos += done.value
        return y if pos == y.shape[0] else y[:pos].copy()  # (y may have grown during the flush)

    def resample_chunk(self, x, last=False):
        """Feed one chunk (1-D mono or 2-D [frame, channel], dtype as constructed); returns the
        output that became available.  Pass last=True exactly once, with the final chunk."""
        if type(x) is not np.ndarray or x.dtype != self._type:
            raise TypeError(
                f"Input should be a `np.ndarray` with matching dtype for ResampleStream({self._type}).")
        if x.ndim == 1:
            return self._process(x, last, True)
        if x.ndim == 2:
            return self._process(x, last)
        raise ValueError("Input must be 1-D or 2-D array")

    def num_clips(self):
        """Number of output samples that saturated (integer I/O)."""
        return int(_n.lib.hipsoxr_stream_num_clips(self._h))

    def delay(self):
        """Pending output, in output samples."""
        return float(_n.lib.hipsoxr_stream_delay(self._h))

    def clear(self):
        """Reset to a fresh signal, keeping the configuration (and the filter bank)."""
        _n.check(_n.lib.hipsoxr_stream_clear(self._h))
        self._ended = False

    def engine(self):
        return _n.lib.hipsoxr_stream_engine(self._h).decode()

    def set_io_ratio(self, in_rate, out_rate, slew_len=0):
        """(Experimental in the reference, src/soxr/__init__.py:162-179.)  New rate ratio for the
        output that follows; needs vr=True.  slew_len > 0: the ratio moves linearly to the new value
        over that many OUTPUT frames; 0: at once.  in_rate/out_rate may not exceed the ratio given
        to the constructor.

        Unit of slew_len: the reference's docstring says "length of smooth transition in input samples"
        (src/soxr/__init__.py:174-177) but passes the number straight to soxr_set_io_ratio
        (src/soxr_ext.cpp:200-204), whose variable-rate stage advances its step once per OUTPUT sample;
        nothing in the reference tests it.  This implementation counts output frames (the engine's
        clock ticks per output, which keeps the position law an exact quadratic: DESIGN.md §3)."""
        if in_rate <= 0 or out_rate <= 0:
            raise ValueError("Sample rate should be over 0")
        _n.check(_n.lib.hipsoxr_stream_set_io_ratio(self._h, float(in_rate) / float(out_rate),
                                                    int(slew_len)))
        # output buffers are sized from the largest out/in ratio seen (src/soxr_ext.cpp:203)
        self._ratio = max(self._ratio, float(out_rate) / float(in_rate))


def _layout_split(x):
    """The reference's dispatch rule (src/soxr/__init__.py:212): unit stride along frames means
    per-channel contiguous ("split") memory."""
    return x.strides[0] == x.itemsize


def _run_oneshot(x2, in_rate, out_rate, recipe, split):
    """x2: 2-D [frame, channel].  One create+process+flush on the device."""
    frames, ch = x2.shape
    elem = _DTYPE_CODES[x2.dtype]
    olen = int(frames * out_rate / in_rate) + 1  # same bound as src/soxr_ext.cpp:238
    done = _C.c_size_t(0)
    if split:
        if frames and x2.strides[0] != x2.itemsize:
            raise ValueError("Data not contiguous")
        buf = np.empty((ch, olen), x2.dtype)  # planar; returned as a Fortran-ordered view
        if frames == 0:
            return buf.T[:0]
        ins = _ptr_array([x2.ctypes.data + c * x2.strides[1] for c in range(ch)])
        outs = _ptr_array([buf.ctypes.data + c * buf.strides[0] for c in range(ch)])
        _n.check(_n.lib.hipsoxr_oneshot(float(in_rate), float(out_rate), ch, ins, frames, outs, olen,
                                        _C.byref(done), elem | 4, recipe, 0))
        return buf.T[:done.value]
    xc = np.ascontiguousarray(x2)
    y = np.empty((olen, ch), x2.dtype)
    if frames == 0:
        return y[:0]
    _n.check(_n.lib.hipsoxr_oneshot(float(in_rate), float(out_rate), ch, xc.ctypes.data, frames,
                                    y.ctypes.data, olen, _C.byref(done), elem, recipe, 0))
    return y[:done.value]


def _run_divided(x2, in_rate, out_rate, recipe, split, div_frames):
    """Long input: one stream handle fed in pieces of `div_frames`, then flushed — the shape of
    csoxr_divide_proc / csoxr_split_ch (src/soxr_ext.cpp:210-273, :277-359)."""
    frames, ch = x2.shape
    elem = _DTYPE_CODES[x2.dtype]
    olen = int(frames * out_rate / in_rate) + 1
    item = x2.itemsize
    h = _C.c_void_p()
    _n.check(_n.lib.hipsoxr_stream_create(float(in_rate), float(out_rate), ch, elem | (4 if split else 0),
                                          recipe, 0, _C.byref(h)))
    try:
        done = _C.c_size_t(0)
        pos = 0
        if split:
            if frames and x2.strides[0] != item:
                raise ValueError("Data not contiguous")
            buf = np.empty((ch, olen), x2.dtype)
            for idx in list(range(0, frames, div_frames)) + [None]:
                n = 0 if idx is None else min(div_frames, frames - idx)
                outs = _ptr_array([buf.ctypes.data + c * buf.strides[0] + pos * item for c in range(ch)])
                ins = None if idx is None else _ptr_array(
                    [x2.ctypes.data + c * x2.strides[1] + idx * item for c in range(ch)])
                _n.check(_n.lib.hipsoxr_stream_process(h, ins, n, outs, olen - pos, _C.byref(done)))
                pos += done.value
            return buf.T[:pos]
        xc = np.ascontiguousarray(x2)
        y = np.empty((olen, ch), x2.dtype)
        row = ch * item
        for idx in list(range(0, frames, div_frames)) + [None]:
            n = 0 if idx is None else min(div_frames, frames - idx)
            ins = None if idx is None else xc.ctypes.data + idx * row
            _n.check(_n.lib.hipsoxr_stream_process(h, ins, n, y.ctypes.data + pos * row, olen - pos,
                                                   _C.byref(done)))
            pos += done.value
        return y[:pos]
    finally:
        _n.lib.hipsoxr_stream_delete(h)


def resample(x, in_rate, out_rate, quality="HQ"):
    """Resample a whole signal.

    x : array_like — mono (1-D) or multi-channel (2-D [frame, channel]); anything that is not an
        ndarray is converted to float32.  dtype float32 / float64 / int16 / int32.
    Returns an ndarray of the same ndim and dtype with floor(frames*out_rate/in_rate + 1/2) frames.
    (Reference: soxr.resample, src/soxr/__init__.py:182-231.)
    """
    _check_rates(in_rate, out_rate)
    if type(x) is not np.ndarray:
        x = np.asarray(x, dtype=np.float32)
    if x.dtype not in _DTYPE_CODES:
        raise TypeError(f"Data type must be one of {_SUPPORTED}, not {x.dtype}")
    recipe = _quality_to_enum(quality)
    if x.ndim == 1:
        x2 = x[:, None]
    elif x.ndim == 2:
        _check_channels(x.shape[1])
        x2 = x
    else:
        raise ValueError("Input must be 1-D or 2-D array")
    # (one channel: split and interleaved memory are the same thing, and the interleaved form is the engine's short path —
    #  pinned host ring, results written straight into host memory: 1000-frame mono call 41 -> 27 us)
    split = _layout_split(x) and x2.shape[1] > 1
    if x2.shape[0] > _DIV_FRAMES:
        y = _run_divided(x2, in_rate, out_rate, recipe, split, _DIV_FRAMES)
    else:
        y = _run_oneshot(x2, in_rate, out_rate, recipe, split)
    return y[:, 0] if x.ndim == 1 else y


def _resample_oneshot(x, in_rate, out_rate, quality="HQ"):
    """Single create+process+flush over a C-contiguous copy (reference: soxr._resample_oneshot,
    src/soxr/__init__.py:234-249 — kept as the test-side cross-check)."""
    if x.dtype not in _DTYPE_CODES:
        raise TypeError(f"Data type must be one of {_SUPPORTED}, not {x.dtype}")
    recipe = _quality_to_enum(quality)
    if x.ndim == 1:
        return _run_oneshot(x[:, None], in_rate, out_rate, recipe, False)[:, 0]
    return _run_oneshot(x, in_rate, out_rate, recipe, False)


def _resample_divided(x, in_rate, out_rate, quality="HQ", div_frames=None):
    """The chunked driver (csoxr_divide_proc / csoxr_split_ch analogue), exposed for tests."""
    recipe = _quality_to_enum(quality)
    x2 = x[:, None] if x.ndim == 1 else x
    if div_frames is None:
        div_frames = max(1000, int(48000 * in_rate / out_rate))  # src/soxr_ext.cpp:239
    y = _run_divided(x2, in_rate, out_rate, recipe, _layout_split(x), int(div_frames))
    return y[:, 0] if x.ndim == 1 else y
