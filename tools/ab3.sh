#!/bin/bash
# Round-3 A/B on one box: [variant|-] [ENV=1 ...] per line of the CONFIGS array; prints headline / batch / configs[2] launch times.
#   tools/ab3.sh "<variant or -> <env assignments>" ...       e.g.  tools/ab3.sh "- " "- HIPSOXR_FFT_NO_PAIR=1" "x2 HIPSOXR_FFT_X2=1"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batch_shard']
print('C1 %.2f us (step %.2f)  batch %.1f us (step %.1f, sustained %.1f)  strong1024 %.0f us  c2 %.1f us  f64arith %.1f us'%(d['roofline']['launch_us'], d['ms_per_step']*1e3, b['roofline']['launch_us'], b['ms_per_step']*1e3, b.get('sustained',{}).get('us_per_launch',0), d.get('batch_strong',{}).get('launch_us_rank0',0), d.get('configs2',{}).get('launch_us',0), d.get('arith_f64',{}).get('launch_us',0)))"; }
CFGS=("$@")
for rep in 1 2; do
for cfg in "${CFGS[@]}"; do
  set -- $cfg; v=$1; shift; envs="$@"
  if [ "$v" = "-" ]; then
    out=$(env $envs python bench.py --no-cpu --steps 60 --windows 20 --sustained-s 1.0 2>/dev/null | tail -1)
  else
    out=$(env $envs tools/with_variant.sh $v python bench.py --no-cpu --steps 60 --windows 20 --sustained-s 1.0 2>/dev/null | tail -1)
  fi
  echo -n "[$v $envs] "; echo "$out" | line
done; done
