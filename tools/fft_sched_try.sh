#!/bin/bash
# tries low-latency (4-pass, prefetched tables) schedules/thread counts for the small paired kernel
# on the 60 s clip with the 900-workgroup switch disabled
sed -i 's/if (wgs < 900) use = &low_latency;/if (wgs < 100000) use = \&low_latency;/' python-soxr_amd/csrc/fft.hip
try() { sed -i "s/typedef PairSpec4<2560, 2352, [0-9]*, 5, 8, 8, 8, [0-9, ]*> Pair2560x2352L;/typedef PairSpec4<2560, 2352, $1, 5, 8, 8, 8, $2> Pair2560x2352L;/" python-soxr_amd/csrc/fft.hip
  bash python-soxr_amd/build.sh > /dev/null 2>&1 || { echo "build failed: $1 $2"; return; }
  echo -n "NT=$1 inv=$2: "; python tools/time_config.py 48000 44100 VHQ 2880000 1 1 5 | grep "^kernel"; }
try 512 "6, 7, 7, 8"
try 384 "7, 7, 8, 6"
try 384 "8, 7, 7, 6"
try 384 "7, 8, 7, 6"
try 448 "7, 7, 8, 6"
try 320 "7, 7, 8, 6"
