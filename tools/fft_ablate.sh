#!/bin/bash
# timing ablations of the paired FFT kernel (results are wrong by construction; timing only)
for abl in 0 1 2 3; do
  HIPSOXR_EXTRA_FLAGS="-DFFT_ABL=$abl" bash python-soxr_amd/build.sh > /dev/null 2>&1
  echo -n "FFT_ABL=$abl (1: no stores, 2: no input loads): "; python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"
done
