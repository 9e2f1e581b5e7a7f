#!/bin/bash
# timing ablations of k_fft_pair2 (results wrong by construction): -DFFT2_ABL bits: 1 = no barriers,
# 2 = no input loads, 4 = no output stores, 8 = no LDS stores in the passes, 16 = no butterflies, 32 = no twiddles
for abl in ${ABLS:-0 1 2 4 6 8 16 32 48 56 62}; do
  HIPSOXR_EXTRA_FLAGS="-DFFT2_ABL=$abl" bash python-soxr_amd/build.sh > /dev/null 2>&1
  echo -n "FFT2_ABL=$abl: "; python bench.py --no-cpu --steps 60 --windows 20 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"
done
bash python-soxr_amd/build.sh > /dev/null 2>&1
