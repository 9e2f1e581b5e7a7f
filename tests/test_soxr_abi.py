"""The libsoxr-named ABI (include/soxr.h, libsoxr.so.0) — SURVEY.md §8(b)(i) / §8(f)-4.

A plain-C client (tests/c/soxr_client.c) is compiled against the header and linked against the
library exactly as a libsoxr user (or the reference's USE_SYSTEM_LIBSOXR build,
/root/reference/CMakeLists.txt:83-93) would, then driven through the call pattern of the
reference binding (src/soxr_ext.cpp:210-273 push loop + flush, :362-402 one-shot, :277-359 split).
Results must be bit-identical to the Python surface, which the parity tests tie to the oracle.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "python-soxr_amd", "soxr_amd")
LIB = os.path.join(LIBDIR, "libsoxr.so.0")
HEADER = os.path.join(ROOT, "include", "soxr.h")


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("soxr_client") / "soxr_client")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "soxr_client.c"), "-o", exe,
                           "-L" + LIBDIR, "-l:libsoxr.so.0", "-Wl,-rpath," + LIBDIR])
    return exe


def _run(exe, *args):
    p = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600)
    return p.returncode, dict(l.split("=", 1) for l in p.stdout.splitlines() if "=" in l), p.stderr


def test_library_exports_every_declared_symbol():
    names = re.findall(r"^SOXR\s[^;]*?\b(soxr_\w+)\s*\(", open(HEADER).read(), flags=re.M)
    assert len(names) == 16 and "soxr_process" in names and "soxr_oneshot" in names
    lib = C.CDLL(LIB)
    for n in names:
        assert hasattr(lib, n), n
    lib.soxr_version.restype = C.c_char_p
    assert b"hipsoxr" in lib.soxr_version()
    # the native ABI is served by the same object
    assert hasattr(lib, "hipsoxr_stream_process")


def test_spec_constructors_by_value(client):
    rc, kv, _ = _run(client, "info")
    assert rc == 0
    assert float(kv["vhq_precision"]) == 28 and abs(float(kv["vhq_passband_end"]) - 0.91151) < 1e-5
    assert kv["io_itype"] == "3" and float(kv["io_scale"]) == 1


def test_spec_struct_layout_via_ctypes():
    class Q(C.Structure):
        _fields_ = [("precision", C.c_double), ("phase_response", C.c_double), ("passband_end", C.c_double),
                    ("stopband_begin", C.c_double), ("e", C.c_void_p), ("flags", C.c_ulong)]
    lib = C.CDLL(LIB)
    lib.soxr_quality_spec.restype = Q
    lib.soxr_quality_spec.argtypes = [C.c_ulong, C.c_ulong]
    for recipe, bits in [(0, 0), (1, 16), (2, 16), (4, 20), (6, 28)]:
        q = lib.soxr_quality_spec(recipe, 32)
        assert (q.precision, q.phase_response, q.stopband_begin, q.flags) == (bits, 50, 1, 32)
    assert lib.soxr_quality_spec(1, 0).passband_end == 1385 / 2048
    assert lib.soxr_quality_spec(4 | 0x30, 0).phase_response == 0   # minimum phase: refused at create


def test_create_reports_errors_through_the_error_pointer():
    lib = C.CDLL(LIB)
    lib.soxr_create.restype = C.c_void_p
    lib.soxr_create.argtypes = [C.c_double, C.c_double, C.c_uint, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p,
                                C.c_void_p]
    err = C.c_char_p()
    h = lib.soxr_create(0.0, 44100.0, 1, C.byref(err), None, None, None)   # bad rate
    assert not h and err.value


DTYPES = {0: np.float32, 1: np.float64, 2: np.int32, 3: np.int16}


@pytest.mark.gpu
@pytest.mark.parametrize("mode,dtype,piece", [("push", 0, 52244), ("push", 1, 1000), ("push", 2, 4410),
                                              ("push", 3, 777), ("push", 4, 52244), ("push", 7, 3001),
                                              ("oneshot", 0, 0), ("oneshot", 3, 0), ("pull", 0, 4096),
                                              ("pull", 3, 500)])
def test_c_client_matches_python_surface(client, soxr, tmp_path, mode, dtype, piece):
    np_t = DTYPES[dtype & 3]
    rng = np.random.default_rng(dtype * 10 + len(mode))
    x = rng.standard_normal((60011, 2))
    x = (x * 5000).astype(np_t) if np.issubdtype(np_t, np.integer) else (x * 0.25).astype(np_t)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    x.tofile(fin)
    in_rate, out_rate, recipe = (48000, 44100, 6) if dtype in (0, 4) else (44100, 16000, 4)
    rc, kv, err = _run(client, mode, in_rate, out_rate, 2, dtype, recipe, piece, fin, fout)
    assert rc == 0, err
    y = np.fromfile(fout, np_t).reshape(-1, 2)
    want = soxr.resample(x, in_rate, out_rate, quality={6: "VHQ", 4: "HQ"}[recipe])
    assert int(kv["frames_out"]) == len(want) == len(y)
    assert np.array_equal(y, want)
    if mode != "oneshot":
        assert "gfx950" in kv["engine"] or "hip" in kv["engine"].lower()
        assert abs(float(kv["delay_end"])) < 1          # everything flushed (fractional remainder only)


@pytest.mark.gpu
def test_c_client_counts_clips(client, tmp_path):
    x = np.full((4000, 1), 32767, np.int16)
    x[::2] = -32768
    x[1000:3000] = 32767
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    x.tofile(fin)
    rc, kv, err = _run(client, "push", 44100, 48000, 1, 3, 4, 1000, fin, fout)
    assert rc == 0, err
    assert int(kv["clips"]) > 0
