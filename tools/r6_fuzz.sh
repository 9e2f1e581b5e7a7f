#!/bin/bash
# the fuzzers on fresh seeds (the wave-kernel fuzzer on the debug-switch build, every eligible job forced) -> gpurun_out/r06_fuzz_extended.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
F=tests/fuzz
{
HIPSOXR_LIBRARY=$R/python-soxr_amd/_variants/dbg/libhipsoxr.so HIPSOXR_DEBUG_WAVE_MIN=1 timeout 900 python $F/fuzz_fft_wave.py 250 6001 2>&1 | tail -2
timeout 900 python $F/fuzz_fft_wave.py 60 6002 2>&1 | tail -1
timeout 900 python $F/fuzz_vs_oracle.py 300 6003 2>&1 | tail -1
timeout 900 python $F/fuzz_device_exact.py 300 6004 2>&1 | tail -1
timeout 900 python $F/fuzz_two_stage.py 200 6005 2>&1 | tail -1
timeout 900 python $F/fuzz_fft_engine.py 200 6006 2>&1 | tail -1
timeout 900 python $F/fuzz_vr.py 300 6007 2>&1 | tail -1
timeout 900 python $F/fuzz_device_stream.py 300 6008 2>&1 | tail -1
timeout 900 python $F/fuzz_stream_group.py 120 6009 2>&1 | tail -3
} > gpurun_out/r06_fuzz_extended.txt 2>&1
cat gpurun_out/r06_fuzz_extended.txt
