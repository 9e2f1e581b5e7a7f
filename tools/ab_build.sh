#!/bin/bash
# A/B of two BUILDS on one box: ab_build.sh "<flags A>" "<flags B>" (HIPSOXR_EXTRA_FLAGS), bench.py headline / batch / configs[2], 3 rounds
for rep in 1 2 3; do
for fl in "$@"; do
  HIPSOXR_EXTRA_FLAGS="$fl" bash python-soxr_amd/build.sh > /dev/null 2>&1
  echo -n "[$fl] "; python bench.py --no-cpu --steps 100 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us (step %.2f)  batch %.1f us  c2 %.1f us'%(d['roofline']['launch_us'], d['ms_per_step']*1e3, d['batch_shard']['roofline']['launch_us'], d.get('configs2',{}).get('launch_us',0)))"
done; done
bash python-soxr_amd/build.sh > /dev/null 2>&1
