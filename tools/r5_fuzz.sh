#!/bin/bash
# the seven fuzzers on fresh seeds -> gpurun_out/r5_fuzz.txt ; then the profile set
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
F=tests/fuzz
{
timeout 900 python $F/fuzz_vs_oracle.py 300 1001 2>&1 | tail -1
timeout 900 python $F/fuzz_device_exact.py 300 1002 2>&1 | tail -1
timeout 900 python $F/fuzz_two_stage.py 200 1003 2>&1 | tail -1
timeout 900 python $F/fuzz_fft_engine.py 200 1004 2>&1 | tail -1
timeout 900 python $F/fuzz_vr.py 300 1005 2>&1 | tail -1
timeout 900 python $F/fuzz_device_stream.py 300 1006 2>&1 | tail -1
timeout 900 python $F/fuzz_stream_group.py 120 1007 2>&1 | tail -3
} > gpurun_out/r5_fuzz.txt 2>&1
cat gpurun_out/r5_fuzz.txt
