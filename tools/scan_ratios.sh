#!/bin/bash
# Kernel time of device-resident jobs over ratios x sizes x engines (exact = 6, AUTO = 0): looks for latency cliffs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for pair in "48000 44100" "44100 48000" "44100 16000" "16000 44100" "48000 16000" "16000 48000" "96000 44100" "44100 96000" "48000 32000" "22050 44100" "48000 8000"; do
  set -- $pair
  for q in ${QUALS:-VHQ HQ}; do
  for f in 20000 200000 2000000; do
    echo -n "$1->$2 $q frames=$f ${DTYPE:-f32}: "; DTYPE=${DTYPE:-f32} python tools/time_config.py $1 $2 $q $f 1 1 0 6 2>&1 | grep "^kernel" | awk '{printf "%s %s us | ", $2, $3}'; echo
  done; done
done
