"""ctypes binding of libhipsoxr.so (include/hipsoxr.h) — the counterpart of the reference's
nanobind module `soxr_ext` (src/soxr_ext.cpp:405-452).

There is no CPU fallback: if the shared library is missing this module raises ImportError, and
every compute entry point returns an error (-> RuntimeError) when no HIP device is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPSOXR_LIBRARY: another build of the same library — the tests and tools that sweep A/B switches load the
# -DHIPSOXR_DEBUG_SWITCHES build (python-soxr_amd/_variants/dbg/) this way; the product build ignores those switches.
LIB_PATH = os.environ.get("HIPSOXR_LIBRARY") or os.path.join(_HERE, "libhipsoxr.so")
if not os.path.exists(LIB_PATH):
    # installed wheel: the one copy of the engine is the libsoxr-named object (it exports both ABIs)
    _alt = os.path.join(_HERE, "prefix", "lib", "libsoxr.so.0")
    if os.path.exists(_alt):
        LIB_PATH = _alt

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first "
        "(python-soxr_amd/build.sh, or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "soxr_amd has no CPU fallback.")

def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 (SONAME
    libamdhip64.so.7, found through RPATH $ORIGIN as `libamdhip64.so`); libhipsoxr.so needs
    `libamdhip64.so.7`.  If the system copy were loaded for us and torch's copy for torch, the
    process would hold two runtimes and only the first to initialise would see the GPU.  Loading
    torch's copy first (when torch is installed) makes both resolve to the same object."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is already loaded; the SONAME match picks it up
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand)
            except OSError:
                pass


_preload_hip_runtime()
lib = C.CDLL(LIB_PATH)

# datatypes (libsoxr numbering)
FLOAT32_I, FLOAT64_I, INT32_I, INT16_I, FLOAT32_S, FLOAT64_S, INT32_S, INT16_S = range(8)
F32, F64, I32, I16 = range(4)
QQ, LQ, MQ, HQ, VHQ = 0, 1, 2, 4, 6
VR = 32
NO_DITHER = 8
DEFER = 64
RESIDENT = 128
AUTO_RESIDENT = 256
KERNEL_AUTO, KERNEL_GATHER, KERNEL_TILE, KERNEL_TILE_VALU, KERNEL_TILE_MFMA, KERNEL_FFT, KERNEL_EXACT, KERNEL_WAVE_DOT, KERNEL_FFT_F64 = range(9)


class PlanInfo(C.Structure):
    _fields_ = [("in_rate", C.c_double), ("out_rate", C.c_double), ("recipe", C.c_ulong),
                ("L", C.c_int64), ("M", C.c_int64), ("taps", C.c_int32), ("interpolated", C.c_int32),
                ("precision_bits", C.c_double), ("passband_end", C.c_double),
                ("stopband_begin", C.c_double), ("att_db", C.c_double), ("kaiser_beta", C.c_double),
                ("bank_elems", C.c_uint64)]


class Job(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("elem", C.c_int32), ("kernel", C.c_int32),
                ("n_clips", C.c_uint32), ("n_channels", C.c_uint32),
                ("in_clip_stride", C.c_int64), ("in_frame_stride", C.c_int64), ("in_chan_stride", C.c_int64),
                ("out_clip_stride", C.c_int64), ("out_frame_stride", C.c_int64), ("out_chan_stride", C.c_int64),
                ("in_abs0", C.c_int64), ("in_frames", C.c_int64),
                ("out_k0", C.c_int64), ("out_frames", C.c_int64),
                ("clip_counter", C.c_void_p), ("dither", C.c_uint32), ("dither_seed", C.c_uint32),
                ("clip_table", C.c_void_p), ("clip_table_dev", C.c_void_p)]


_err = C.c_char_p
_P = C.POINTER

# Every symbol include/hipsoxr.h declares (tests/test_cabi.py checks the list against the header).
SIGNATURES = {
    "hipsoxr_version": (C.c_char_p, []),
    "hipsoxr_device_count": (C.c_int, []),
    "hipsoxr_plan_create": (_err, [C.c_double, C.c_double, C.c_ulong, _P(C.c_void_p)]),
    "hipsoxr_plan_create_vr": (_err, [C.c_double, C.c_double, C.c_ulong, _P(C.c_void_p)]),
    "hipsoxr_plan_delete": (None, [C.c_void_p]),
    "hipsoxr_plan_info": (_err, [C.c_void_p, _P(PlanInfo)]),
    "hipsoxr_plan_get_bank": (_err, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hipsoxr_plan_set_bank": (_err, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hipsoxr_plan_broadcast": (_err, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hipsoxr_plan_out_len": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "hipsoxr_run_device": (_err, [C.c_void_p, _P(Job), C.c_void_p]),
    "hipsoxr_stream_create": (_err, [C.c_double, C.c_double, C.c_uint, C.c_int, C.c_ulong, C.c_ulong,
                                     _P(C.c_void_p)]),
    "hipsoxr_stream_create_with_plan": (_err, [C.c_void_p, C.c_uint, C.c_int, C.c_ulong, _P(C.c_void_p)]),
    "hipsoxr_stream_process": (_err, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      _P(C.c_size_t)]),
    "hipsoxr_stream_process_device": (_err, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             _P(C.c_size_t), C.c_void_p]),
    "hipsoxr_streams_process_device": (_err, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hipsoxr_stream_delete": (None, [C.c_void_p]),
    "hipsoxr_stream_clear": (_err, [C.c_void_p]),
    "hipsoxr_stream_delay": (C.c_double, [C.c_void_p]),
    "hipsoxr_stream_num_clips": (C.c_size_t, [C.c_void_p]),
    "hipsoxr_stream_engine": (C.c_char_p, [C.c_void_p]),
    "hipsoxr_stream_set_io_ratio": (_err, [C.c_void_p, C.c_double, C.c_size_t]),
    "hipsoxr_stream_plan": (C.c_void_p, [C.c_void_p]),
    "hipsoxr_stream_set_dither_seed": (_err, [C.c_void_p, C.c_uint32]),
    "hipsoxr_oneshot": (_err, [C.c_double, C.c_double, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p,
                               C.c_size_t, _P(C.c_size_t), C.c_int, C.c_ulong, C.c_ulong]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)  # AttributeError here == the library does not export the ABI
    _f.restype = _res
    _f.argtypes = _args


def check(err):
    """libsoxr convention: non-NULL error string -> RuntimeError (src/soxr_ext.cpp:80-82)."""
    if err:
        raise RuntimeError(err.decode() if isinstance(err, bytes) else str(err))


def version():
    return lib.hipsoxr_version().decode()


# The ctypes mirrors above (Job = hipsoxr_job_t with its clip_table fields) are laid out for this ABI generation: a
# library of another generation would read garbage from the struct's tail, so loading one is an import error.
ABI_VERSION = "hipsoxr-0.5"
if not version().startswith(ABI_VERSION):
    raise ImportError(f"{LIB_PATH} is {version()!r}; this package binds {ABI_VERSION}.x (rebuild: python-soxr_amd/build.sh)")


def device_count():
    return int(lib.hipsoxr_device_count())
