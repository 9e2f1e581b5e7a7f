#!/bin/bash
# Batch-only A/B on one box: each argument is a set of env assignments ("-" = none); 3 interleaved rounds of
# tools/run_workload.py batch 200 (us per launch over 200 back-to-back launches).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
CFGS=("$@")
for rep in 1 2 3; do
for cfg in "${CFGS[@]}"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  echo -n "[$cfg] "; env $e python tools/run_workload.py ${WORKLOAD:-batch} ${LAUNCHES:-200} 2>/dev/null | tail -n 1
done; done
