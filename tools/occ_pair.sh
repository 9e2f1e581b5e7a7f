#!/bin/bash
# occupancy sensitivity of the paired FFT kernel: default LDS (40 KB/workgroup) vs 54 KB (3/CU) vs 80 KB (2/CU),
# full kernel and compute-only ablation (FFT_ABL=3: no loads, no stores; results wrong by construction)
for abl in 0 3; do
  HIPSOXR_EXTRA_FLAGS="-DFFT_ABL=$abl" bash python-soxr_amd/build.sh > /dev/null 2>&1
  for lds in 0 54000 81920; do
    echo -n "FFT_ABL=$abl LDS=$lds: "; HIPSOXR_DEBUG_FFT_LDS=$lds python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"
  done
done
bash python-soxr_amd/build.sh > /dev/null 2>&1
