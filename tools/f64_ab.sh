#!/bin/bash
# float64-engine device jobs (int32 and float64 I/O, exact engine = kernel 6) on the configs[1] shape and the configs[2]
# ratio, with and without the f64 MFMA tile kernel, and across its slab sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "-" "HIPSOXR_NO_MFMA64=1" "HIPSOXR_DEBUG_MFMA64_LDS=60000" "HIPSOXR_DEBUG_MFMA64_LDS=30000"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  for w in i32 f64; do echo -n "[$cfg] "; env $e python tools/run_workload.py $w 50 6 2>/dev/null | tail -n 1; done
  echo -n "[$cfg] 44.1k->16k int32 60 s mono: "; env $e DTYPE=i32 python tools/time_config.py 44100 16000 VHQ 2646000 1 1 6 2>/dev/null | tail -n 1
done
