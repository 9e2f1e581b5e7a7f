#!/bin/bash
# timing of FFT-engine variants on the bench workloads (debug switches; not part of the product path)
run() { echo -n "$1: "; env $2 python bench.py --no-cpu --steps 100 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"; }
run default ""
run no_pair HIPSOXR_FFT_NO_PAIR=1
run large_only HIPSOXR_FFT_LARGE_ONLY=1
run large_only_no_pair "HIPSOXR_FFT_LARGE_ONLY=1 HIPSOXR_FFT_NO_PAIR=1"
