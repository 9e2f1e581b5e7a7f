#!/bin/bash
# General MFMA kernel (k_tile_mfma, 44.1k -> 16k VHQ int16 mono): default rule vs 16-period slabs + half-chains forced
# (HIPSOXR_DEBUG_SLAB32=1) vs 64-period slabs (HIPSOXR_DEBUG_SLAB64=1), job sizes in 64-period slabs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for slabs in ${SLABS:-50 100 127 150 200 300 500 800 1500 3000}; do
  frames=$((slabs * 64 * 441))
  for e in "A=1" "HIPSOXR_DEBUG_SLAB32=1" "HIPSOXR_DEBUG_SLAB32=1 HIPSOXR_DEBUG_NO_HALVES=1" "HIPSOXR_DEBUG_SLAB64=1"; do
    echo -n "slabs64=$slabs [$e]: "; env $e DTYPE=i16 python tools/time_config.py 44100 16000 VHQ $frames 1 1 6 2>&1 | tail -1 | cut -c1-30
  done
done
