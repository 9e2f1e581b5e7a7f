#!/bin/bash
# round 6: float32 I/O on float64 arithmetic (HIPSOXR_KERNEL_FFT_F64) on the batch shard, by block size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DBG=$PWD/python-soxr_amd/_variants/dbg/libhipsoxr.so
{
for rep in 1 2; do
echo "== 1280 x 1176 (default)"; KERNEL=8 HIPSOXR_LIBRARY=$DBG timeout 300 python tools/wave_check.py 128 10 | tail -2
echo "== 2560 x 2352 (HIPSOXR_FFT_NO_TINY)"; KERNEL=8 HIPSOXR_FFT_NO_TINY=1 HIPSOXR_LIBRARY=$DBG timeout 300 python tools/wave_check.py 128 10 | tail -2
done
} > gpurun_out/r6_f64.txt 2>&1
cat gpurun_out/r6_f64.txt
