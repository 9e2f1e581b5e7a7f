#!/bin/bash
# A variant of the library that differs from the product in csrc/fftwave.hip's build alone (and reads the debug switches):
#   tools/wave_variant.sh <name> [extra hipcc flags...]   ->  python-soxr_amd/_variants/<name>/libhipsoxr.so
# Needs the product's objects (python-soxr_amd/_obj, made by build.sh).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)/python-soxr_amd"
N=$1; shift
mkdir -p $R/_obj/$N $R/_variants/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I$R/../include \
  -ffp-contract=fast -fno-slp-vectorize -DHIPSOXR_DEBUG_SWITCHES "$@" -save-temps=obj -c $R/csrc/fftwave.hip -o $R/_obj/$N/fftwave.o
O=$R/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/plan.o $O/engine.o $O/kernels_dbg.o $O/fft.o $O/fft1.o $O/fft2.o $O/twostage.o $O/$N/fftwave.o $O/soxr_abi.o \
  -o $R/_variants/$N/libhipsoxr.so
S=$O/$N/fftwave-hip-amdgcn-amd-amdhsa-gfx950.s
python $R/../tools/isa_stats.py $S 'k_fft_wave.*3840ELi3528' | head -4
grep -A12 "amdhsa_kernel.*3840ELi3528" $S | grep -E "private_segment|next_free_vgpr" ; grep -E "vgpr_spill_count" $S | head -2
