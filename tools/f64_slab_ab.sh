#!/bin/bash
# float64 planar MFMA kernel (k_tile_mfma64_p; int32 I/O, exact engine): slab size (HIPSOXR_DEBUG_MFMA64_PB=16/32),
# 16-period units on 32-period slabs (HIPSOXR_DEBUG_MFMA64_SPLIT) and the unit split over job sizes in 64-period slabs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for slabs in ${SLABS:-10 47 141 282 564 1128 3008}; do
  frames=$((slabs * 64 * 160))
  for e in "A=1" "HIPSOXR_DEBUG_MFMA64_PB=16" "HIPSOXR_DEBUG_MFMA64_PB=32" "HIPSOXR_DEBUG_MFMA64_PB=32 HIPSOXR_DEBUG_MFMA64_SPLIT=1" "HIPSOXR_DEBUG_MFMA64_PB=16 HIPSOXR_DEBUG_SPLIT=3" "HIPSOXR_DEBUG_MFMA64_PB=32 HIPSOXR_DEBUG_MFMA64_SPLIT=1 HIPSOXR_DEBUG_SPLIT=5" "HIPSOXR_DEBUG_MFMA64_PB=16 HIPSOXR_DEBUG_NW=5"; do
    echo -n "slabs64=$slabs [$e]: "; env $e DTYPE=i32 python tools/time_config.py 48000 44100 VHQ $frames 1 1 6 2>&1 | tail -1 | cut -c1-30
  done
done
