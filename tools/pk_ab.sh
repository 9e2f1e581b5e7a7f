#!/bin/bash
# packed-f32 complex arithmetic in the FFT kernels (variant build -DFFT_PK) against the product: parity tests, then timing
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/with_variant.sh pk python -m pytest tests/test_gpu_fft.py tests/test_gpu_frequency_response.py -q -x 2>&1 | tail -2
for rep in 1 2 3; do
  for w in batch clip c2 f64; do
    echo -n "[product] $w: "; ROTATE=4 python tools/run_workload.py $w 300 2>&1 | tail -n 1
    echo -n "[pk]      $w: "; ROTATE=4 tools/with_variant.sh pk python tools/run_workload.py $w 300 2>&1 | tail -n 1
  done
done
