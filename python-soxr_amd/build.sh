#!/bin/bash
# Build libhipsoxr.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
#
# Two flag sets:
#   * everything on the canonical-order path (plan design, exact-engine kernels) is compiled with
#     -ffp-contract=off: every fused multiply-add there is written explicitly, so that host design
#     and device arithmetic are bit-identical to the oracle's;
#   * the frequency-domain engine (fft.hip) is a 1e-6-class path: contraction is allowed, and the
#     SLP vectoriser is off — on gfx950 v_pk_{add,mul,fma}_f32 issue at half the rate of their
#     scalar forms, so "vectorised" complex arithmetic only adds register shuffles
#     (measured: 5412 -> 3261 VALU issue slots per thread for the 2560/2352-point block).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
# HIPSOXR_VARIANT=<name>: an experiment build (with HIPSOXR_EXTRA_FLAGS) into _variants/<name>/ beside the product
# library, so that several builds made here travel to the GPU box in one snapshot (tools/with_variant.sh swaps one in).
OUTDIR="$HERE/soxr_amd"
OBJ="$HERE/_obj"   # (not build/: that name belongs to setuptools when a wheel is built from this directory)
if [ -n "$HIPSOXR_VARIANT" ]; then OUTDIR="$HERE/_variants/$HIPSOXR_VARIANT"; OBJ="$HERE/_obj/$HIPSOXR_VARIANT"; mkdir -p "$OUTDIR"; fi
OUT="$OUTDIR/libhipsoxr.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I$HERE/../include"
mkdir -p "$OBJ"
"$HIPCC" $COMMON -ffp-contract=off ${HIPSOXR_EXTRA_FLAGS} -c "$SRC/plan.cpp" -o "$OBJ/plan.o" & P1=$!
"$HIPCC" $COMMON -ffp-contract=off ${HIPSOXR_EXTRA_FLAGS} -c "$SRC/engine.cpp" -o "$OBJ/engine.o" & P2=$!
"$HIPCC" $COMMON -ffp-contract=off ${HIPSOXR_EXTRA_FLAGS} -c "$SRC/kernels.hip" -o "$OBJ/kernels.o" & P3=$!
"$HIPCC" $COMMON -ffp-contract=off ${HIPSOXR_EXTRA_FLAGS} -c "$SRC/soxr_abi.cpp" -o "$OBJ/soxr_abi.o" & P5=$!
# The A/B and timing-experiment switches (device.h `Switches`, everything but four product names) are read from the
# environment only by a build with -DHIPSOXR_DEBUG_SWITCHES.  The reader lives in kernels.hip alone, so the debug build is
# that one object compiled a second time and linked with the product's other objects: _variants/dbg/libhipsoxr.so
# (tests/test_gpu_switches.py, tests/test_gpu_launch_forms.py and tools/*.sh load it through HIPSOXR_LIBRARY).
P8=""
if [ -z "$HIPSOXR_VARIANT" ]; then
  "$HIPCC" $COMMON -ffp-contract=off ${HIPSOXR_EXTRA_FLAGS} -DHIPSOXR_DEBUG_SWITCHES -c "$SRC/kernels.hip" -o "$OBJ/kernels_dbg.o" & P8=$!
fi
# (fft.hip in three translation units: see "Three translation units" there)
FFTFLAGS="${HIPSOXR_FFTFLAGS:--ffp-contract=fast -fno-slp-vectorize}"
"$HIPCC" $COMMON $FFTFLAGS ${HIPSOXR_EXTRA_FLAGS} -DFFT_PART=0 -c "$SRC/fft.hip" -o "$OBJ/fft.o" & P4=$!
"$HIPCC" $COMMON $FFTFLAGS ${HIPSOXR_EXTRA_FLAGS} -DFFT_PART=1 -c "$SRC/fft.hip" -o "$OBJ/fft1.o" & P6=$!
"$HIPCC" $COMMON $FFTFLAGS ${HIPSOXR_EXTRA_FLAGS} -DFFT_PART=2 -c "$SRC/fft.hip" -o "$OBJ/fft2.o" & P7=$!
"$HIPCC" $COMMON $FFTFLAGS ${HIPSOXR_EXTRA_FLAGS} -c "$SRC/twostage.hip" -o "$OBJ/twostage.o" & P9=$!
"$HIPCC" $COMMON $FFTFLAGS ${HIPSOXR_EXTRA_FLAGS} -c "$SRC/fftwave.hip" -o "$OBJ/fftwave.o" & P10=$!
wait $P1; wait $P2; wait $P3; wait $P4; wait $P5; wait $P6; wait $P7; wait $P9; wait $P10   # set -e: any failed compile aborts here
OBJS="$OBJ/plan.o $OBJ/engine.o $OBJ/kernels.o $OBJ/fft.o $OBJ/fft1.o $OBJ/fft2.o $OBJ/twostage.o $OBJ/fftwave.o $OBJ/soxr_abi.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
# The same engine under libsoxr's name: what `find_library(SOXR_LIBRARY NAMES soxr)` of the
# reference's USE_SYSTEM_LIBSOXR build picks up (reference CMakeLists.txt:83-93).
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libsoxr.so.0 $OBJS -o "$OUTDIR/libsoxr.so.0"
ln -sf libsoxr.so.0 "$OUTDIR/libsoxr.so"
if [ -n "$P8" ]; then
  wait $P8
  mkdir -p "$HERE/_variants/dbg"
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC ${OBJS/kernels.o/kernels_dbg.o} -o "$HERE/_variants/dbg/libhipsoxr.so"
fi
echo "built $OUT (+ libsoxr.so.0)"
