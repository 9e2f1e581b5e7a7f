for l in 0 22000 24000 27000 30000 36000 48000 64000; do echo -n "lds=$l "; HIPSOXR_DEBUG_FFT_LDS=$l python bench.py --no-cpu --steps 50 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"; done
