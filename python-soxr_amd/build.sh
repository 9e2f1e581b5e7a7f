#!/bin/bash
# Build libhipsoxr.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/soxr_amd/libhipsoxr.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function"
"$HIPCC" $FLAGS ${HIPSOXR_EXTRA_FLAGS} -I"$HERE/../include" \
    "$SRC/plan.cpp" "$SRC/engine.cpp" "$SRC/kernels.hip" "$SRC/fft.hip" -o "$OUT"
echo "built $OUT"
