"""ctypes front-end of the CPU oracle (oracle/soxr_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package (python-soxr_amd/).  See the header of soxr_oracle.c for what it restates and the
parity status ("parity unpinned" against libsoxr itself; pinned against the reference's
known-answer tests).

Two halves: the filter DESIGN is oracle/design.py (numpy/scipy, a formulation deliberately unlike the
product's plan.cpp); the ARITHMETIC, given a bank, is soxr_oracle.c.  Which bank a comparison
uses decides what it tests:
  mode="ref"  always runs on the oracle's OWN bank (design.py): GPU-vs-ref comparisons (<= 1e-6)
              therefore check design and arithmetic against independently written code;
  mode="port" checks the ORDER of the arithmetic bit for bit, which is only meaningful on identical
              coefficients: when `bank_provider` is set (tests/conftest.py, smoke(), installed from
              the product's C ABI: hipsoxr_plan_get_bank) port mode runs on the product's float64
              bank; the two banks themselves are compared in tests/test_design_independent.py.

`resample(x, in_rate, out_rate, quality, mode)` mirrors soxr.resample's array contract
(/root/reference/src/soxr/__init__.py:182-231): 1-D or 2-D [frame, channel] input of dtype
float32/float64/int16/int32, same ndim/dtype out, floor(n*out/in + 1/2) frames.
  mode="port": canonical order in the engine precision (f32 engine for float32/int16 I/O,
               f64 engine for float64/int32 I/O) — the HIP path must match this BIT FOR BIT.
  mode="ref" : float64 accumulation; returned as float64 before any output rounding.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import design

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

QQ, LQ, MQ, HQ, VHQ = 0, 1, 2, 4, 6
_QUALITY = {"qq": QQ, "lq": LQ, "mq": MQ, "hq": HQ, "vhq": VHQ,
            "soxr_qq": QQ, "soxr_lq": LQ, "soxr_mq": MQ, "soxr_hq": HQ, "soxr_vhq": VHQ}


def build(force=False):
    """Compile liboracle.so with gcc (a few hundred lines of C; ~1 s)."""
    src = os.path.join(_HERE, "soxr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        i64, i32, u32, u64, dbl = C.c_int64, C.c_int32, C.c_uint32, C.c_uint64, C.c_double
        P = C.POINTER
        L.oracle_out_len.argtypes = [u64, i64, i64]
        L.oracle_out_len.restype = u64
        for name in ("oracle_resample_ref", "oracle_resample_port_f64", "oracle_resample_port_f32"):
            getattr(L, name).argtypes = [C.c_void_p, i64, i64, i32, C.c_void_p, i64, i64,
                                         C.c_void_p, i64, i64]
            getattr(L, name).restype = None
        for name in ("oracle_interp_ref", "oracle_interp_port_f64", "oracle_interp_port_f32"):
            getattr(L, name).argtypes = [C.c_void_p, i32, i64, i64, i32, C.c_void_p, i64, i64,
                                         C.c_void_p, i64, i64]
            getattr(L, name).restype = None
        for name in ("oracle_vr_ref", "oracle_vr_port_f64", "oracle_vr_port_f32"):
            getattr(L, name).argtypes = [C.c_void_p, i32, i32, C.c_void_p, i64, i64, C.c_void_p, i64] + [u64] * 6
            getattr(L, name).restype = None
        L.oracle_dither.argtypes = [u32, u32, i64]
        L.oracle_dither.restype = dbl
        L.oracle_quantize_i16.argtypes = [C.c_void_p, i64, C.c_int, u32, u32, i64, C.c_void_p]
        L.oracle_quantize_i16.restype = u64
        L.oracle_quantize_i32.argtypes = [C.c_void_p, i64, C.c_void_p]
        L.oracle_quantize_i32.restype = u64
        _lib = L
    return _lib


def quality_enum(q):
    if isinstance(q, str):
        return _QUALITY[q.lower()]
    return int(q)


def quality(recipe):
    return design.quality(quality_enum(recipe))


# Optional source of the PRODUCT's float64 bank for port-mode (bit-exact arithmetic-order) checks:
# callable(in_rate, out_rate, recipe, vr) -> ndarray, installed by tests/conftest.py and smoke().
bank_provider = None


class Plan:
    """Geometry + float64 bank for (in_rate, out_rate, recipe), designed by oracle/design.py:
    [L][T] for an exact plan (phases == 0), else the interpolated-phase table [P][T][4]."""

    def __init__(self, in_rate, out_rate, recipe="HQ"):
        self.in_rate, self.out_rate, self.recipe = float(in_rate), float(out_rate), quality_enum(recipe)
        g = design.geometry(self.in_rate, self.out_rate, self.recipe)
        self.L, self.M, self.T, self.att_db, self.beta = g["L"], g["M"], g["T"], g["att_db"], g["beta"]
        self.phases = g["phases"]
        if self.phases:
            self.bank = design.interp_table(self.in_rate, self.out_rate, self.recipe)
        else:
            self.bank = design.bank(self.in_rate, self.out_rate, self.recipe)
        self._port_bank = None

    @property
    def port_bank(self):
        """Bank for port-mode runs: the product's when a provider is installed, else our own."""
        if bank_provider is None:
            return self.bank
        if self._port_bank is None:
            b = np.ascontiguousarray(bank_provider(self.in_rate, self.out_rate, self.recipe, False), np.float64)
            assert b.shape == self.bank.shape, (b.shape, self.bank.shape)
            self._port_bank = b
        return self._port_bank

    def exact_coefs(self, f):
        """Un-interpolated coefficients c_j(f) of an interpolated-phase plan (accuracy checks)."""
        return design.exact_coefs(self.in_rate, self.out_rate, self.recipe, f)

    def out_len(self, n_in):
        return int(lib().oracle_out_len(int(n_in), self.L, self.M))


_plans = {}


def plan(in_rate, out_rate, recipe="HQ"):
    key = (float(in_rate), float(out_rate), quality_enum(recipe))
    if key not in _plans:
        _plans[key] = Plan(*key)
    return _plans[key]


def resample_channel(pl, x, mode, k0=0, n_out=None, in_abs0=0, bank=None):
    """One planar channel.  x: float32 (port f32), float64 (port f64 / ref).  Returns the engine
    output before integer quantisation."""
    if bank is None:
        bank = pl.bank if mode == "ref" else pl.port_bank
    bank = np.ascontiguousarray(bank, np.float64)
    if n_out is None:
        n_out = pl.out_len(len(x)) - k0
    if mode == "ref":
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty(n_out, np.float64)
        fn = lib().oracle_interp_ref if pl.phases else lib().oracle_resample_ref
    elif mode == "port_f64":
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty(n_out, np.float64)
        fn = lib().oracle_interp_port_f64 if pl.phases else lib().oracle_resample_port_f64
    elif mode == "port_f32":
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(n_out, np.float32)
        fn = lib().oracle_interp_port_f32 if pl.phases else lib().oracle_resample_port_f32
    else:
        raise ValueError(mode)
    if pl.phases:
        fn(bank.ctypes.data, pl.phases, pl.L, pl.M, pl.T, x.ctypes.data, in_abs0, len(x), y.ctypes.data,
           k0, n_out)
    else:
        fn(bank.ctypes.data, pl.L, pl.M, pl.T, x.ctypes.data, in_abs0, len(x), y.ctypes.data, k0, n_out)
    return y


class VrPlan:
    """The table of a variable-rate stream created with (in_rate, out_rate) = the largest io ratio:
    always an interpolated-phase table, whatever the ratio."""

    def __init__(self, in_rate, out_rate, recipe="HQ"):
        r = quality_enum(recipe)
        self.T = design.geometry(float(in_rate), float(out_rate), r)["T"]
        self.phases = design.vr_phases(r)
        self.bank = design.interp_table(float(in_rate), float(out_rate), r, self.phases)
        self.port_bank = self.bank
        if bank_provider is not None:
            b = np.ascontiguousarray(bank_provider(float(in_rate), float(out_rate), r, True), np.float64)
            assert b.shape == self.bank.shape, (b.shape, self.bank.shape)
            self.port_bank = b


def vr_run(vp, x, mode, n_out, T0, S0, D, in_abs0=0):
    """Outputs i < n_out at Q64.64 positions T0 + i*S0 + D*i(i-1)/2 (Python integers; D may be
    negative).  mode: "ref" | "port_f64" | "port_f32"."""
    real = np.float32 if mode == "port_f32" else np.float64
    x = np.ascontiguousarray(x, real)
    y = np.empty(n_out, real)
    words = []
    for v in (T0, S0, D):
        v &= (1 << 128) - 1
        words += [v >> 64, v & ((1 << 64) - 1)]
    bank = vp.bank if mode == "ref" else vp.port_bank
    getattr(lib(), "oracle_vr_" + mode)(bank.ctypes.data, vp.phases, vp.T, x.ctypes.data, in_abs0, len(x),
                                        y.ctypes.data, n_out, *words)
    return y


def engine_of(dtype):
    """Engine precision the product uses for an I/O dtype: f32 for float32/int16, f64 otherwise."""
    dtype = np.dtype(dtype)
    return "f32" if dtype in (np.dtype(np.float32), np.dtype(np.int16)) else "f64"


def quantize(v, dtype, channel=0, k0=0, dither=True, seed=0):
    """Engine output -> I/O dtype, returning (array, n_clips)."""
    dtype = np.dtype(dtype)
    if dtype == np.int16:
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty(len(v), np.int16)
        clips = lib().oracle_quantize_i16(v.ctypes.data, len(v), int(bool(dither)), seed, channel, k0,
                                          out.ctypes.data)
        return out, int(clips)
    if dtype == np.int32:
        v = np.ascontiguousarray(v, np.float64)
        out = np.empty(len(v), np.int32)
        clips = lib().oracle_quantize_i32(v.ctypes.data, len(v), out.ctypes.data)
        return out, int(clips)
    return np.asarray(v).astype(dtype, copy=False), 0


def resample(x, in_rate, out_rate, quality="HQ", mode="port", dither=True, seed=0,
             return_clips=False):
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float64, np.int16, np.int32):
        raise TypeError(x.dtype)
    pl = plan(in_rate, out_rate, quality)
    squeeze = x.ndim == 1
    x2 = x[:, None] if squeeze else x
    n_out = pl.out_len(x2.shape[0])
    eng = engine_of(x.dtype)
    cols, clips = [], 0
    for c in range(x2.shape[1]):
        xc = x2[:, c]
        if mode == "ref":
            cols.append(resample_channel(pl, xc.astype(np.float64), "ref", n_out=n_out))
        else:
            real = np.float32 if eng == "f32" else np.float64
            v = resample_channel(pl, xc.astype(real), "port_" + eng, n_out=n_out)
            q, nc = quantize(v, x.dtype, channel=c, dither=dither, seed=seed)
            cols.append(q)
            clips += nc
    y = np.stack(cols, axis=1) if cols else np.empty((n_out, 0), x.dtype)
    y = y[:, 0] if squeeze else y
    return (y, clips) if return_clips else y
