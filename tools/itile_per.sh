#!/bin/bash
# k_interp_tile: the largest mean bucket the launcher may pick (HIPSOXR_DEBUG_ITILE_PER; above 64 a row walk serves two sets of
# outputs) on exact-engine jobs of interpolated-phase plans.  Needs the debug-switch build.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HIPSOXR_LIBRARY=python-soxr_amd/_variants/dbg/libhipsoxr.so
for per in 64 80 96 112 128; do
  echo "== per <= $per"
  HIPSOXR_DEBUG_ITILE_PER=$per python tools/time_config.py 48000 44101 VHQ 2880000 2 1 6 2>&1 | grep "kernel 6"
  HIPSOXR_DEBUG_ITILE_PER=$per python tools/time_config.py 48000 44101 VHQ 2880000 1 1 6 2>&1 | grep "kernel 6"
  HIPSOXR_DEBUG_ITILE_PER=$per DTYPE=i16 python tools/time_config.py 44100 16001 VHQ 2646000 2 1 6 2>&1 | grep "kernel 6"
  HIPSOXR_DEBUG_ITILE_PER=$per python tools/time_config.py 48000 44101 HQ 480000 2 1 6 2>&1 | grep "kernel 6"
done
