#!/bin/bash
# board power per VALU form (tools/ubench/valu_power.hip): runs each mode while sampling rocm-smi
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in 6 0 1 2 3 4 5; do
  tools/ubench/valu_power $m > /tmp/vp.out 2>&1 &
  pid=$!
  sleep 1.2
  pw=""; sc=""
  while kill -0 $pid 2>/dev/null; do
    o=$(rocm-smi --showpower --showclocks 2>/dev/null)
    pw="$pw $(echo "$o" | grep -oE 'Power \(W\): [0-9.]+' | grep -oE '[0-9.]+$')"
    sc="$sc $(echo "$o" | grep -oE 'sclk clock level: [0-9]+: \([0-9]+Mhz' | grep -oE '[0-9]+Mhz' | head -1)"
  done
  wait $pid
  echo "$(cat /tmp/vp.out) | power:$pw | sclk:$sc"
done
