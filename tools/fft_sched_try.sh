#!/bin/bash
# tries inverse schedules for the small paired kernel (single 60 s clip: latency-bound)
try() { sed -i "s/typedef PairSpec<2560, 2352, [0-9]*, 16, 16, 10, true, [0-9, a-z]*> Pair2560x2352;/typedef PairSpec<2560, 2352, $1, 16, 16, 10, true, $2> Pair2560x2352;/" python-soxr_amd/csrc/fft.hip
  bash python-soxr_amd/build.sh > /dev/null 2>&1 || { echo "build failed: $1 $2"; return; }
  echo -n "NT=$1 inv=$2: "; python tools/time_config.py 48000 44100 VHQ 2880000 1 1 0 | grep "^kernel"; }
try 384 "21, 16, 7, false"
try 384 "14, 12, 14, false"
try 384 "7, 16, 21, false"
try 384 "12, 14, 14, false"
try 256 "14, 12, 14, false"
