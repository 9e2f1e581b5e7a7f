#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of two-stage jobs, k_poly2 against k_poly (debug build switch)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
export HIPSOXR_LIBRARY=$R/python-soxr_amd/_variants/dbg/libhipsoxr.so
cd /tmp && export TMPDIR=/tmp
{
for v in pair single; do
  if [ $v = single ]; then export HIPSOXR_POLY_NO_PAIR=1; else unset HIPSOXR_POLY_NO_PAIR; fi
  for cfg in "48000 44101 VHQ 2880000 2" "44101 48000 VHQ 2880000 2" "44100 16001 VHQ 2880000 2" "48000 44101 VHQ 2880000 1"; do
    rm -rf /tmp/tsp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/tsp -o t -- python $R/tools/two_stage_prof.py $cfg 30 > /tmp/tsp.log 2>&1
    echo "== [$v] $cfg"; python $R/tools/pmc_summary.py /tmp/tsp/*.db 2>/dev/null | grep -v "at::native\|rocclr\|per-grid" | grep "k_poly\|k_fft\|k_tile\|k_interp" | head -4
  done
done
} 2>&1 | tee $R/gpurun_out/r5_poly2_prof.txt
