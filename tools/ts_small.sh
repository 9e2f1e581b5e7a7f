#!/bin/bash
# two-stage job: the FFT stage on full-size vs half-size blocks (debug build, HIPSOXR_FFT_SMALL_ONLY)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export HIPSOXR_LIBRARY=$R/python-soxr_amd/_variants/dbg/libhipsoxr.so
for env in "A=1" "HIPSOXR_FFT_SMALL_ONLY=1"; do
for cfg in "48000 44101 VHQ 2880000 2" "44101 48000 VHQ 2880000 2" "48000 44101 VHQ 2880000 1" "48000 44101 VHQ 960000 2"; do
  rm -rf /tmp/tsp; env $env timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/tsp -o t -- python $R/tools/two_stage_prof.py $cfg 30 > /tmp/tsp.log 2>&1
  echo "== $env $cfg"; python $R/tools/pmc_summary.py /tmp/tsp/*.db 2>/dev/null | grep -v "at::native\|rocclr\|per-grid" | grep "k_poly\|k_fft" | head -3 | cut -c1-110
done; done
