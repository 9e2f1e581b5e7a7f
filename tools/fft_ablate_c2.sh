#!/bin/bash
# timing ablations of the FIRST-generation paired kernel in channel-pair mode on configs[2] (results wrong by
# construction; HIPSOXR_FFT_PAIR_V1 selects that kernel, AUTO now runs k_fft_strided2)
for abl in 0 1 2 3; do
  HIPSOXR_EXTRA_FLAGS="-DFFT_ABL=$abl" bash python-soxr_amd/build.sh > /dev/null 2>&1
  echo -n "FFT_ABL=$abl (1: no stores, 2: no input loads): "; HIPSOXR_FFT_PAIR_V1=1 python bench.py --no-cpu --steps 50 --batch-clips 8 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 %.1f us'%(d['configs2']['launch_us']))"
done
bash python-soxr_amd/build.sh > /dev/null 2>&1
