"""`python -m soxr_amd --prefix` prints the install prefix of the libsoxr-named ABI (lib/libsoxr.so,
include/soxr.h, lib/pkgconfig/soxr.pc): what CMAKE_PREFIX_PATH / PKG_CONFIG_PATH of a libsoxr client —
e.g. the reference's USE_SYSTEM_LIBSOXR build, CMakeLists.txt:83-93 — should point at."""
import os
import sys

from . import prefix


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    p = prefix()
    if argv[:1] == ["--pkgconfig"]:
        print(os.path.join(p, "lib", "pkgconfig"))
    elif argv[:1] in ([], ["--prefix"]):
        print(p)
    else:
        print("usage: python -m soxr_amd [--prefix | --pkgconfig]", file=sys.stderr)
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
