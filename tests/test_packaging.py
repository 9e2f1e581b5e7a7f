"""wheel packaging around the libsoxr-named ABI.

Builds the wheel from python-soxr_amd/ (setuptools, no network, the in-tree library is reused when it is
up to date), installs it into an empty directory, and — in a subprocess that cannot see the source
tree — checks that the package imports, that `python -m soxr_amd --prefix` points at a prefix in which
the reference's own way of finding libsoxr works (CMakeLists.txt:83-93: find_library(NAMES soxr) and
find_path(soxr.h) below CMAKE_PREFIX_PATH; here: the same layout, plus pkg-config), and that a plain-C
libsoxr client compiles, links and runs against it.  The GPU variant then drives the client through
the reference binding's call patterns against the INSTALLED library.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "python-soxr_amd")


@pytest.fixture(scope="module")
def installed(tmp_path_factory):
    work = tmp_path_factory.mktemp("wheel")
    site = str(work / "site")
    subprocess.check_call([sys.executable, "-m", "pip", "wheel", "--no-build-isolation", "--no-deps", "--no-index",
                           "-q", "-w", str(work), SRC], cwd=str(work))
    whl = [f for f in os.listdir(work) if f.endswith(".whl")]
    assert len(whl) == 1 and whl[0].startswith("soxr_amd-")
    subprocess.check_call([sys.executable, "-m", "pip", "install", "--no-deps", "--no-index", "-q", "--target", site,
                           str(work / whl[0])])
    return site


def _py(site, code):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = site
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=site, timeout=300)


def test_wheel_installs_and_imports_without_the_source_tree(installed):
    p = _py(installed, "import soxr_amd, soxr, os; print(soxr_amd.__file__); print(soxr_amd.__libsoxr_version__); "
                       "print(soxr.resample is soxr_amd.resample); print(soxr_amd.prefix())")
    assert p.returncode == 0, p.stderr
    f, ver, same, prefix = p.stdout.split("\n")[:4]
    assert f.startswith(installed) and ver.startswith("hipsoxr") and same == "True"
    for rel in ("lib/libsoxr.so", "lib/libsoxr.so.0", "include/soxr.h", "include/hipsoxr.h", "lib/pkgconfig/soxr.pc"):
        assert os.path.exists(os.path.join(prefix, rel)), rel


def test_one_version_number_everywhere(installed):
    """include/hipsoxr.h's HIPSOXR_VERSION_STRING == the native library's report == soxr_amd.__version__ == the wheel's
    metadata == soxr.pc (hygiene: three strings used to disagree)."""
    import re
    hdr = re.search(r'#define\s+HIPSOXR_VERSION_STRING\s+"([^"]+)"', open(os.path.join(ROOT, "include", "hipsoxr.h")).read()).group(1)
    p = _py(installed, "import soxr_amd, importlib.metadata as md; print(soxr_amd.__version__); print(soxr_amd.__libsoxr_version__); "
                       "print(md.version('soxr-amd'))")
    assert p.returncode == 0, p.stderr
    pyver, native, wheel = p.stdout.split("\n")[:3]
    assert pyver == hdr and wheel == hdr and native == "hipsoxr-%s (gfx950)" % hdr, (hdr, pyver, native, wheel)
    pc = open(os.path.join(installed, "soxr_amd", "prefix", "lib", "pkgconfig", "soxr.pc")).read()
    assert "Version: %s\n" % hdr in pc
    whl_dirs = [d for d in os.listdir(installed) if d.endswith(".dist-info")]
    assert whl_dirs == ["soxr_amd-%s.dist-info" % hdr], whl_dirs


@pytest.fixture(scope="module")
def client(installed, tmp_path_factory):
    """tests/c/soxr_client.c built the way a libsoxr user would: header and library found below the
    installed prefix (what find_library / find_path do with CMAKE_PREFIX_PATH)."""
    prefix = os.path.join(installed, "soxr_amd", "prefix")
    exe = str(tmp_path_factory.mktemp("client") / "soxr_client")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(prefix, "include"),
                           os.path.join(ROOT, "tests", "c", "soxr_client.c"), "-o", exe,
                           "-L" + os.path.join(prefix, "lib"), "-lsoxr", "-Wl,-rpath," + os.path.join(prefix, "lib")])
    return exe


def test_c_client_links_against_installed_prefix(client):
    p = subprocess.run([client, "info"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    kv = dict(l.split("=", 1) for l in p.stdout.splitlines() if "=" in l)
    assert float(kv["vhq_precision"]) == 28 and "hipsoxr" in kv.get("version", "hipsoxr")


def test_pkgconfig_file_is_relocatable(installed):
    pc = open(os.path.join(installed, "soxr_amd", "prefix", "lib", "pkgconfig", "soxr.pc")).read()
    assert "${pcfiledir}" in pc and "-lsoxr" in pc and "Name: soxr" in pc


@pytest.mark.gpu
def test_installed_library_runs_the_reference_call_patterns(installed, client, tmp_path):
    """The installed libsoxr.so through a C client (push loop + flush, as csoxr_divide_proc does:
    /root/reference/src/soxr_ext.cpp:210-273) == the installed Python surface, bit for bit."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((30000, 2)) * 0.25).astype(np.float32)
    fin, fout = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    x.tofile(fin)
    # MODE in_rate out_rate channels dtype recipe piece infile outfile
    p = subprocess.run([client, "push", "48000", "44100", "2", "0", "6", "4800", fin, fout], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.fromfile(fout, np.float32).reshape(-1, 2)
    q = _py(installed, "import numpy as np, soxr_amd as soxr; x=np.fromfile(%r,np.float32).reshape(-1,2); "
                       "soxr.resample(x,48000,44100,quality='VHQ').tofile(%r)" % (fin, fout + ".py"))
    assert q.returncode == 0, q.stderr
    want = np.fromfile(fout + ".py", np.float32).reshape(-1, 2)
    assert got.shape == want.shape and np.array_equal(got, want)
