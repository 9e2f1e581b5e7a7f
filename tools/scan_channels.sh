#!/bin/bash
# Device-resident multi-channel / multi-clip jobs over ratios x sizes (AUTO = 0, EXACT = 6): kernel time, us
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for pair in "48000 44100" "44100 48000" "44100 16000" "16000 48000"; do
  set -- $pair
  for shape in "2 1" "8 1" "1 16" "2 64"; do
    set -- $pair $shape
    for f in 20000 200000 2000000; do
      echo -n "$1->$2 VHQ ch=$3 clips=$4 frames=$f ${DTYPE:-f32}: "; DTYPE=${DTYPE:-f32} python tools/time_config.py $1 $2 VHQ $f $3 $4 0 6 2>&1 | grep "^kernel" | awk '{printf "%s %s us | ", $2, $3}'; echo
    done
  done
done
