#!/bin/bash
# Exact engine (kernel 6, float32 arithmetic): slab size (32 / 64 periods) and unit split over job sizes given in
# 64-period slabs (48k -> 44.1k VHQ mono; a 60 s clip is 282).  Default = launch_tile's cost model;
# HIPSOXR_DEBUG_SLAB64=1 = round 2's rule (64-period slabs, split min(5, 1536 / slabs) below 512 slabs);
# HIPSOXR_DEBUG_SLAB32=1 / HIPSOXR_DEBUG_SPLIT=n force a form.      tools/slab_ab.sh  [MODE=sweep]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
if [ "$MODE" = "sweep" ]; then
  CF=("HIPSOXR_DEBUG_SLAB64=1 HIPSOXR_DEBUG_SPLIT=1" "HIPSOXR_DEBUG_SLAB64=1 HIPSOXR_DEBUG_SPLIT=5" "HIPSOXR_DEBUG_SLAB32=1 HIPSOXR_DEBUG_SPLIT=1" "HIPSOXR_DEBUG_SLAB32=1 HIPSOXR_DEBUG_SPLIT=3")
else
  CF=("A=1" "HIPSOXR_DEBUG_SLAB64=1")
fi
for slabs in ${SLABS:-5 10 20 32 40 47 64 80 100 141 200 256 282 330 376 450 511 520 600 768 1024 1500}; do
  frames=$((slabs * 64 * 160))
  for e in "${CF[@]}"; do
    echo -n "slabs64=$slabs [$e]: "; env $e python tools/time_config.py 48000 44100 VHQ $frames 1 1 6 2>&1 | tail -1 | cut -c1-30
  done
done
