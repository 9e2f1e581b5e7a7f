"""The C-ABI library loads and exports every symbol include/hipsoxr.h declares (no compute)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hipsoxr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"HIPSOXR_API[^;(]*?\b(hipsoxr_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("hipsoxr_stream_create", "hipsoxr_stream_process", "hipsoxr_stream_delete",
                 "hipsoxr_stream_clear", "hipsoxr_stream_delay", "hipsoxr_stream_num_clips",
                 "hipsoxr_stream_engine", "hipsoxr_stream_set_io_ratio", "hipsoxr_oneshot",
                 "hipsoxr_version", "hipsoxr_run_device", "hipsoxr_plan_create"):
        assert must in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from soxr_amd import _native
    out = subprocess.check_output(["nm", "-D", "--defined-only", _native.LIB_PATH], text=True)
    exported = set(re.findall(r" T (hipsoxr_\w+)", out))
    for s in declared_symbols():
        assert s in exported, f"{s} declared in include/hipsoxr.h but not exported"
        assert hasattr(_native.lib, s)
        assert s in _native.SIGNATURES, f"{s} has no ctypes signature in _native.SIGNATURES"
    assert exported == set(declared_symbols()), "library exports symbols the header does not declare"


def test_no_oracle_in_product():
    """The shipped package must not reach into oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "python-soxr_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_version_and_device_count_do_not_need_a_gpu():
    from soxr_amd import _native
    assert _native.version().startswith("hipsoxr-")
    assert _native.device_count() >= 0


def test_product_library_reads_four_environment_names():
    """The product build reads HIPSOXR_NO_FFT, HIPSOXR_RESIDENT, HIPSOXR_AUTO_RESIDENT, HIPSOXR_RESIDENT_IDLE_US and no
    other HIPSOXR_* name: the A/B and timing-experiment switches exist only in the -DHIPSOXR_DEBUG_SWITCHES build
    (python-soxr_amd/_variants/dbg/), which tests/test_gpu_switches.py sweeps."""
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "python-soxr_amd")

    def names(path):
        with open(path, "rb") as f:
            blob = f.read()
        return set(m.decode() for m in re.findall(rb"HIPSOXR_[A-Z0-9_]{3,}", blob))

    prod = names(os.path.join(pkg, "soxr_amd", "libhipsoxr.so"))
    allowed = {"HIPSOXR_NO_FFT", "HIPSOXR_RESIDENT", "HIPSOXR_AUTO_RESIDENT", "HIPSOXR_RESIDENT_IDLE_US"}
    env_like = {n for n in prod if not n.startswith(("HIPSOXR_KERNEL_", "HIPSOXR_F32", "HIPSOXR_F64", "HIPSOXR_I"))}
    assert env_like <= allowed, sorted(env_like - allowed)
    dbg = os.path.join(pkg, "_variants", "dbg", "libhipsoxr.so")
    assert os.path.exists(dbg)
    assert {"HIPSOXR_FFT_NO_PAIR", "HIPSOXR_DEBUG_SLAB64", "HIPSOXR_NO_PLANES"} <= names(dbg)
