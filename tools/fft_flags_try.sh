#!/bin/bash
# compiler-flag experiments on the frequency-domain engine (timing only)
try() { HIPSOXR_EXTRA_FLAGS="$1" bash python-soxr_amd/build.sh > /dev/null 2>&1 || { echo "build failed: $1"; return; }
  echo -n "[$1]: "; python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"; }
try ""
try "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"
try "-mllvm -amdgpu-schedule-metric-bias=0"
try "-mllvm -amdgpu-schedule-metric-bias=100"
try "-mllvm -enable-post-misched=0"
try "-mllvm -amdgpu-use-aa-in-codegen=0"
try ""
