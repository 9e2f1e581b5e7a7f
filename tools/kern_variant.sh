#!/bin/bash
# A variant of the library that differs from the product in csrc/kernels.hip's build alone (debug switches on):
#   tools/kern_variant.sh <name> [extra hipcc flags...]   ->  python-soxr_amd/_variants/<name>/libhipsoxr.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)/python-soxr_amd"
N=$1; shift
mkdir -p $R/_obj/$N $R/_variants/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I$R/../include \
  -ffp-contract=off -DHIPSOXR_DEBUG_SWITCHES "$@" -c $R/csrc/kernels.hip -o $R/_obj/$N/kernels.o
O=$R/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/plan.o $O/engine.o $O/$N/kernels.o $O/fft.o $O/fft1.o $O/fft2.o $O/twostage.o $O/fftwave.o $O/soxr_abi.o \
  -o $R/_variants/$N/libhipsoxr.so
echo built $N
