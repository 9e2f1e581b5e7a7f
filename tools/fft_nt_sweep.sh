#!/bin/bash
# rebuilds fft.hip with different thread counts for the small paired kernel and times both workloads with it
for nt in 192 256 384; do
  sed -i "s/static constexpr int NA = 2560, NB = 2352, NT = [0-9]*;/static constexpr int NA = 2560, NB = 2352, NT = $nt;/" python-soxr_amd/csrc/fft.hip
  bash python-soxr_amd/build.sh > /dev/null 2>&1
  echo -n "small NT=$nt: "; HIPSOXR_FFT_SMALL_ONLY=1 python bench.py --no-cpu --steps 100 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1 %.2f us  batch %.1f us'%(d['roofline']['launch_us'], d['batch_shard']['roofline']['launch_us']))"
done
