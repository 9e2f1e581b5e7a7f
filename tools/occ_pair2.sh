#!/bin/bash
for lds in 0 41472 45056 51200 54272 54784 61440 81920; do
  echo -n "LDS=$lds: "; HIPSOXR_DEBUG_FFT_LDS=$lds python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"
done
