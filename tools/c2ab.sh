# configs[2] (8 ch interleaved) under environment switches: c2ab.sh "VAR=x" ... ("-" = none)
for e in "$@"; do
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  echo -n "[$e] "; env $envs python bench.py --no-cpu --batch-clips 8 --steps 50 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 %.1f us frac %.3f'%(d.get('configs2',{}).get('launch_us',0), d.get('configs2',{}).get('roofline',{}).get('frac',0)))"
done
