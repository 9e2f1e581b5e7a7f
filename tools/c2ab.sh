for S in 60 600; do for e in - HIPSOXR_FFT_SMALL_ONLY=1; do
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  echo -n "[$S s $e] "; env $envs python bench.py --no-cpu --batch-clips 8 --steps 50 --seconds $S 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  c2 %.1f us frac %.3f'%(d['roofline']['launch_us'], d.get('configs2',{}).get('launch_us',0), d.get('configs2',{}).get('roofline',{}).get('frac',0)))"
done; done
