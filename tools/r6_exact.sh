#!/bin/bash
# round 6: the exact engine (k_tile_mfma_p) on the 60 s mono clip and on int16 / int32, per library variant:  tools/r6_exact.sh <variant>...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
V=$PWD/python-soxr_amd/_variants
{
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"
  HIPSOXR_LIBRARY=$V/$v/libhipsoxr.so timeout 300 python tools/time_config.py 48000 44100 VHQ 2880000 1 1 6 2>&1 | grep "kernel 6"
  DTYPE=i16 HIPSOXR_LIBRARY=$V/$v/libhipsoxr.so timeout 300 python tools/time_config.py 48000 44100 VHQ 2880000 1 1 6 2>&1 | grep "kernel 6"
  HIPSOXR_LIBRARY=$V/$v/libhipsoxr.so timeout 300 python tools/time_config.py 48000 44100 VHQ 480000 1 128 6 2>&1 | grep "kernel 6"
done; done
} > gpurun_out/r6_exact.txt 2>&1
cat gpurun_out/r6_exact.txt
