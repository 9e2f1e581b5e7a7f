"""setuptools glue: compile the HIP library with build.sh (hipcc, gfx950) when it is missing or stale,
then lay the libsoxr-named ABI out as an install prefix inside the package (see pyproject.toml)."""
import os
import shutil
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "soxr_amd")
INCLUDE = os.path.join(HERE, "..", "include")



def header_version():
    """The ONE version number: HIPSOXR_VERSION_STRING in include/hipsoxr.h (what hipsoxr_version() / soxr_version() report,
    what soxr_amd.__version__ is derived from at import, and what the wheel and soxr.pc carry)."""
    import re
    with open(os.path.join(INCLUDE, "hipsoxr.h")) as f:
        m = re.search(r'#define\s+HIPSOXR_VERSION_STRING\s+"([0-9][^"]*)"', f.read())
    if not m:
        raise RuntimeError("HIPSOXR_VERSION_STRING not found in include/hipsoxr.h")
    return m.group(1)


VERSION = header_version()

PC = """prefix=${pcfiledir}/../..
libdir=${prefix}/lib
includedir=${prefix}/include

Name: soxr
Description: libsoxr-compatible ABI of the MI355X (gfx950) resampler (hipsoxr)
Version: @VERSION@
Libs: -L${libdir} -lsoxr -Wl,-rpath,${libdir}
Cflags: -I${includedir}
"""


def _stale(lib):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    src = os.path.join(HERE, "csrc")
    return any(os.path.getmtime(os.path.join(src, f)) > t for f in os.listdir(src))


class build_with_hip(build_py):
    def run(self):
        lib = os.path.join(PKG, "libhipsoxr.so")
        if _stale(lib):
            subprocess.check_call(["bash", os.path.join(HERE, "build.sh")])
        prefix = os.path.join(PKG, "prefix")
        shutil.rmtree(prefix, ignore_errors=True)
        os.makedirs(os.path.join(prefix, "lib", "pkgconfig"))
        os.makedirs(os.path.join(prefix, "include"))
        # one copy of the engine: libsoxr.so.0 (SONAME libsoxr.so.0) serves both ABIs; the Python surface loads it too
        shutil.copy2(os.path.join(PKG, "libsoxr.so.0"), os.path.join(prefix, "lib", "libsoxr.so.0"))
        shutil.copy2(os.path.join(PKG, "libsoxr.so.0"), os.path.join(prefix, "lib", "libsoxr.so"))  # (wheels hold no symlinks)
        for h in ("soxr.h", "hipsoxr.h"):
            shutil.copy2(os.path.join(INCLUDE, h), os.path.join(prefix, "include", h))
        with open(os.path.join(prefix, "lib", "pkgconfig", "soxr.pc"), "w") as f:
            f.write(PC.replace("@VERSION@", VERSION))
        super().run()


setup(
    name="soxr-amd",
    version=VERSION,
    description="MI355X (gfx950) implementation of python-soxr's resampling hot path: soxr.resample / "
                "ResampleStream over hand-written HIP kernels",
    python_requires=">=3.9",
    install_requires=["numpy"],
    entry_points={"console_scripts": ["soxr-amd-prefix=soxr_amd.__main__:main"]},
    packages=["soxr_amd", "soxr"],
    package_data={"soxr_amd": ["prefix/lib/libsoxr.so.0", "prefix/lib/libsoxr.so", "prefix/lib/pkgconfig/soxr.pc",
                               "prefix/include/*.h"]},
    cmdclass={"build_py": build_with_hip},
)
