#!/bin/bash
# non-temporal result stores (product) against plain stores (variant nont: -DFFT_STORE_AUX=0; the exact-engine half of this experiment, -DHIPSOXR_EXACT_STORE_NT, is no longer in the source), every workload
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for w in "batch 200 0" "clip 300 0" "c2 200 0" "f64 200 0" "batch 60 6" "clip 200 6" "i32 200 0" "c2 100 6"; do
  for v in nt nont; do
    echo -n "[$v] [$w] "
    if [ $v = nt ]; then python tools/run_workload.py $w 2>&1 | tail -n 1; else tools/with_variant.sh nont python tools/run_workload.py $w 2>&1 | tail -n 1; fi
  done
done; done
echo "== host API"
python tools/time_host_api.py 2>&1 | tail -8
tools/with_variant.sh nont python tools/time_host_api.py 2>&1 | tail -8
