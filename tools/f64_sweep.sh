#!/bin/bash
# float64 MFMA tile kernel: waves per workgroup x slab size, int32 60 s mono 48k -> 44.1k (exact engine)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lds in 120000 30000; do for nw in 0 5 4 2; do
  e="HIPSOXR_DEBUG_MFMA64_LDS=$lds"; [ $nw != 0 ] && e="$e HIPSOXR_DEBUG_NW=$nw"
  echo -n "[$e] "; env $e python tools/run_workload.py i32 50 6 2>/dev/null | tail -n 1
done; done
