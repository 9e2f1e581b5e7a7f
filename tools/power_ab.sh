#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showpower --showclocks 2>&1 | head -30
S="HIPSOXR_FFT_SMALL_ONLY=1 HIPSOXR_FFT_NO_TINY=1"
for cfg in "HIPSOXR_FFT_X2=0" "HIPSOXR_FFT_X2=0 ZERO_INPUT=1" "HIPSOXR_FFT_X2=0 $S HIPSOXR_DEBUG_NW=4" "HIPSOXR_FFT_X2=1 $S" "HIPSOXR_FFT_X2=1" "KERNEL=6"; do
  echo -n "[$cfg] "; env $cfg tools/with_variant.sh ntsweep python tools/power_probe.py batch 4 2>&1 | tail -n 2
done
