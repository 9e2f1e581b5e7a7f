#!/bin/bash
# round 6: the frequency-domain engine on a 128-clip mono batch (~10 s each) across the standard ratios: where is it weakest?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for r in "48000 44100" "44100 48000" "96000 48000" "48000 96000" "48000 16000" "16000 48000" "48000 32000" "32000 48000" "44100 16000" "16000 44100" "88200 48000" "96000 44100" "44100 32000" "48000 8000"; do
  set -- $r; n=$(( $1 * 10 )); [ $n -gt 500000 ] && n=$(( $1 * 5 ))
  echo "== $1 -> $2, 128 x $n frames"; timeout 300 python tools/time_config.py $1 $2 VHQ $n 1 128 5 2>&1 | grep "kernel 5"
done
} > gpurun_out/r6_ratios.txt 2>&1
cat gpurun_out/r6_ratios.txt
