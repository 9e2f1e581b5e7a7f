cd ${GRAFT_REPO_ROOT:-/root/repo}
S="HIPSOXR_FFT_SMALL_ONLY=1 HIPSOXR_FFT_NO_TINY=1"
for rep in 1 2; do
for cfg in "HIPSOXR_FFT_X2=0" "HIPSOXR_FFT_X2=0 ZERO_INPUT=1" "HIPSOXR_FFT_X2=0 $S HIPSOXR_DEBUG_NW=4" "HIPSOXR_FFT_X2=0 $S HIPSOXR_DEBUG_NW=4 ZERO_INPUT=1" "HIPSOXR_FFT_X2=1 $S" "HIPSOXR_FFT_X2=1 $S ZERO_INPUT=1"; do
  echo -n "[$cfg] "; env $cfg tools/with_variant.sh ntsweep python tools/run_workload.py batch 300 2>&1 | tail -n 1
done; done
