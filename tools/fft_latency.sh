#!/bin/bash
# single-workgroup latency of the paired FFT kernel: tiny jobs, with and without halves of the work
for abl in 0 4 6 3; do
  HIPSOXR_EXTRA_FLAGS="-DFFT_ABL=$abl" bash python-soxr_amd/build.sh > /dev/null 2>&1
  echo "FFT_ABL=$abl (1 no stores, 2 no input loads, 4 forward only)"
  for n in 24000 240000 2880000; do python tools/time_config.py 48000 44100 VHQ $n 1 1 5 2>&1 | grep "^kernel"; done
done
