#!/bin/bash
# batch / clip A/B of the two-pairs-per-workgroup kernel on one box (debug-switch build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export HIPSOXR_LIBRARY=$R/python-soxr_amd/_variants/ntsweep/libhipsoxr.so
for rep in 1 2 3; do
for cfg in "HIPSOXR_FFT_X2=0" "HIPSOXR_FFT_X2=1" "HIPSOXR_FFT_X2=1 HIPSOXR_FFT_SMALL_ONLY=1 HIPSOXR_FFT_NO_TINY=1" "HIPSOXR_FFT_X2=0 HIPSOXR_FFT_SMALL_ONLY=1 HIPSOXR_FFT_NO_TINY=1"; do
  echo -n "[$cfg] "; env $cfg python tools/run_workload.py batch 200 2>&1 | tail -n 1
done; done
for cfg in "HIPSOXR_FFT_X2=0" "HIPSOXR_FFT_X2=1"; do
  echo -n "[$cfg] "; env $cfg python tools/run_workload.py clip 200 2>&1 | tail -n 1
done
