#!/bin/bash
# configs[2] HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, KiB) of the product build and of a variant: tools/c2_traffic.sh <variant>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/c2p_$c -o p -- python $R/tools/run_workload.py c2 6 > /tmp/c2p.log 2>&1
  $R/tools/with_variant.sh $V rocprofv3 --kernel-trace --pmc $c -d /tmp/c2v_$c -o p -- python $R/tools/run_workload.py c2 6 > /tmp/c2v.log 2>&1
done
cd $R
for w in p v; do echo "== $w"; python tools/pmc_summary.py $(find /tmp/c2${w}_FETCH_SIZE /tmp/c2${w}_WRITE_SIZE -name "*.db") 2>&1 | grep -A2 "'k_fft_strided2" | grep "SIZE"; done
