#!/bin/bash
# round 6: packed FMAs in the workgroup FFT kernels on the latency-regime jobs (60 s mono, configs[2], 60 s stereo two-stage)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
V=$PWD/python-soxr_amd/_variants
{
for rep in 1 2 3; do
for v in dbg pkall; do
echo "== $v"
HIPSOXR_LIBRARY=$V/$v/libhipsoxr.so timeout 300 python tools/time_config.py 48000 44100 VHQ 2880000 1 1 5 2>&1 | grep "kernel 5"
HIPSOXR_LIBRARY=$V/$v/libhipsoxr.so timeout 300 python tools/time_config.py 44100 16000 VHQ 2646000 8 1 5 2>&1 | grep "kernel 5"
HIPSOXR_LIBRARY=$V/$v/libhipsoxr.so timeout 300 python tools/time_config.py 48000 44101 VHQ 2880000 2 1 0 2>&1 | grep "kernel 0"
done; done
} > gpurun_out/r6_pk_lat.txt 2>&1
cat gpurun_out/r6_pk_lat.txt
