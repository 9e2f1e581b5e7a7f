#!/bin/bash
# A/B of bench.py under environment switches: ab.sh "VAR1=x" "VAR2=y" ... ("-" = no switch); 3 repeats each
for rep in 1 2 3; do
for e in "$@"; do
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  echo -n "[$e] "; env $envs python bench.py --no-cpu --steps 100 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us (step %.2f)  batch %.1f us  c2 %.1f us'%(d['roofline']['launch_us'], d['ms_per_step']*1e3, d['batch_shard']['roofline']['launch_us'], d.get('configs2',{}).get('launch_us',0)))"
done; done
