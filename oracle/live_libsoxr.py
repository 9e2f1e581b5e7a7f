"""Run-time probe for a LIVE libsoxr (the real reference engine).  TEST INFRASTRUCTURE ONLY.

SURVEY.md §0.2 / §8d(4): libsoxr is absent from this image and from the GPU box (same image), so
parity with libsoxr itself is unpinned.  Should one ever be present — python-soxr installed in
site-packages, or a system libsoxr.so — this module finds it, so that
tests/test_live_libsoxr.py can assert the north-star bar against the real thing and bench.py can
time it as `cpu_baseline.kind == "reference"`.  It never resolves to this repository's own
libsoxr-named ABI (python-soxr_amd/soxr_amd/libsoxr.so.0) or its `soxr` alias package.

    probe() -> None | Live   with  .resample(x, in_rate, out_rate, quality) , .version , .how
"""
import ctypes as C
import ctypes.util
import importlib.machinery
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
_RECIPE = {"qq": 0, "lq": 1, "mq": 2, "hq": 4, "vhq": 6}
_DTYPE = {"float32": 0, "float64": 1, "int32": 2, "int16": 3}   # SOXR_*_I (reference: src/soxr_ext.cpp:35-46)


class _IoSpec(C.Structure):
    _fields_ = [("itype", C.c_int), ("otype", C.c_int), ("scale", C.c_double), ("e", C.c_void_p),
                ("flags", C.c_ulong)]


class _QSpec(C.Structure):
    _fields_ = [("precision", C.c_double), ("phase_response", C.c_double), ("passband_end", C.c_double),
                ("stopband_begin", C.c_double), ("e", C.c_void_p), ("flags", C.c_ulong)]


class Live:
    def __init__(self, how, version, fn):
        self.how, self.version, self.resample = how, version, fn


def _ours(path):
    return bool(path) and os.path.realpath(path).startswith(os.path.realpath(_REPO) + os.sep)


def _probe_python():
    """python-soxr itself (the reference package), looked up on every sys.path entry that is not this
    repository — `import soxr` here would find our own alias package first."""
    paths = [p for p in sys.path if p and not _ours(os.path.join(p, "x"))]
    spec = importlib.machinery.PathFinder.find_spec("soxr", paths)
    if spec is None or spec.origin is None or _ours(spec.origin):
        return None
    mod = importlib.util.module_from_spec(spec)
    saved = sys.modules.get("soxr")
    sys.modules["soxr"] = mod
    try:
        spec.loader.exec_module(mod)
        ver = str(getattr(mod, "__libsoxr_version__", "?"))
        if ver.startswith("hipsoxr"):
            return None
        return Live(f"python package soxr {getattr(mod, '__version__', '?')} at {spec.origin}", ver,
                    lambda x, i, o, q="HQ": mod.resample(x, i, o, quality=q))
    except Exception:
        return None
    finally:
        if saved is not None:
            sys.modules["soxr"] = saved
        else:
            sys.modules.pop("soxr", None)


def _probe_ctypes():
    """A system libsoxr through its C API, called the way the reference's csoxr_oneshot does
    (/root/reference/src/soxr_ext.cpp:375-389: soxr_io_spec, soxr_quality_spec, soxr_oneshot)."""
    name = ctypes.util.find_library("soxr")
    if not name:
        return None
    try:
        lib = C.CDLL(name)
        lib.soxr_version.restype = C.c_char_p
        ver = lib.soxr_version().decode()
        if ver.startswith("hipsoxr"):
            return None
        lib.soxr_io_spec.restype, lib.soxr_io_spec.argtypes = _IoSpec, [C.c_int, C.c_int]
        lib.soxr_quality_spec.restype, lib.soxr_quality_spec.argtypes = _QSpec, [C.c_ulong, C.c_ulong]
        lib.soxr_oneshot.restype = C.c_char_p
        lib.soxr_oneshot.argtypes = [C.c_double, C.c_double, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_IoSpec),
                                     C.POINTER(_QSpec), C.c_void_p]
    except (OSError, AttributeError):
        return None

    def resample(x, in_rate, out_rate, quality="HQ"):
        x = np.ascontiguousarray(x)
        x2 = x[:, None] if x.ndim == 1 else x
        io = lib.soxr_io_spec(_DTYPE[x.dtype.name], _DTYPE[x.dtype.name])
        qs = lib.soxr_quality_spec(_RECIPE[quality.lower()] if isinstance(quality, str) else int(quality), 0)
        olen = int(x2.shape[0] * out_rate / in_rate + 1)
        y = np.zeros((olen, x2.shape[1]), x.dtype)
        odone = C.c_size_t()
        err = lib.soxr_oneshot(in_rate, out_rate, x2.shape[1], x2.ctypes.data, x2.shape[0], None,
                               y.ctypes.data, olen, C.byref(odone), C.byref(io), C.byref(qs), None)
        if err:
            raise RuntimeError(err.decode())
        y = y[:odone.value]
        return y[:, 0] if x.ndim == 1 else y

    return Live(f"system library {name} via ctypes", ver, resample)


_cached = False
_live = None


def probe():
    global _cached, _live
    if not _cached:
        _live = _probe_python() or _probe_ctypes()
        _cached = True
    return _live
