#!/bin/bash
# round 5, first GPU call: FFT parity on the new build, then base (round-4 kernels) vs new, interleaved on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_size.py tests/test_gpu_frequency_response.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5_ab1_tests.txt
B=$R/python-soxr_amd/_variants/base/libhipsoxr.so
for rep in 1 2 3; do
  for w in batch clip c2; do
    for v in base new; do
      if [ $v = base ]; then export HIPSOXR_LIBRARY=$B; else unset HIPSOXR_LIBRARY; fi
      echo -n "[$v rot3] "; ROTATE=3 python tools/run_workload.py $w 300 2>/dev/null | tail -n 1
    done
  done
done > gpurun_out/r5_ab1.txt 2>&1
unset HIPSOXR_LIBRARY
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err
tail -c 600 gpurun_out/r5_bench1.err
cat gpurun_out/r5_ab1_tests.txt gpurun_out/r5_ab1.txt
