cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for cfg in "A=1" "HIPSOXR_DEBUG_NW=3" "HIPSOXR_DEBUG_NW=4" "HIPSOXR_DEBUG_NW=5" "HIPSOXR_FFT_LARGE_ONLY=1" "HIPSOXR_FFT_LARGE_ONLY=1 HIPSOXR_DEBUG_NW=4" "HIPSOXR_FFT_X2=1"; do
  echo -n "[$cfg] "; env $cfg tools/with_variant.sh ntsweep python tools/run_workload.py clip 400 2>&1 | tail -n 1
done; done
