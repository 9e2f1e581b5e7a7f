#!/bin/bash
# thread-count sweep of the 48k -> 44.1k frequency-domain kernels (variant build ntsweep: -DFFT_NT_SWEEP -DFFT_EXPERIMENT_X2 -DHIPSOXR_DEBUG_SWITCHES)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
S="HIPSOXR_FFT_SMALL_ONLY=1 HIPSOXR_FFT_NO_TINY=1"
for rep in 1 2; do
for cfg in "HIPSOXR_FFT_X2=0" "HIPSOXR_FFT_X2=0 HIPSOXR_DEBUG_NW=5" "HIPSOXR_FFT_X2=0 HIPSOXR_DEBUG_NW=4" \
           "HIPSOXR_FFT_X2=0 $S" "HIPSOXR_FFT_X2=0 $S HIPSOXR_DEBUG_NW=5" "HIPSOXR_FFT_X2=0 $S HIPSOXR_DEBUG_NW=4" "HIPSOXR_FFT_X2=0 $S HIPSOXR_DEBUG_NW=3" \
           "HIPSOXR_FFT_X2=1 $S" "HIPSOXR_FFT_X2=1 $S HIPSOXR_DEBUG_NW=3" "HIPSOXR_FFT_X2=1 $S HIPSOXR_DEBUG_NW=6"; do
  echo -n "[$cfg] "; env $cfg tools/with_variant.sh ntsweep python tools/run_workload.py batch 200 2>&1 | tail -n 1
done; done
