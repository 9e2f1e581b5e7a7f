#!/bin/bash
# round 6 A/B on one box: tools/r6_ab.sh <variant>[@nowave] ...   ("trace" = per-wave stamps; @nowave = HIPSOXR_FFT_NO_WAVE=1: k_fft_pair2)
# (HIPSOXR_DEBUG_WAVE_MIN=1: the wave kernel whatever the product's size rule says)
# CLIPS / SECS / RATES pick the job (default 128 x 10 s, 48000 44100)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NOCHECK=${NOCHECK:-}
V=$PWD/python-soxr_amd/_variants
{
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = trace ]; then [ $rep = 1 ] && HIPSOXR_LIBRARY=$V/trace/libhipsoxr.so timeout 300 python tools/trace_wave.py; continue; fi
  name=${v%@nowave}; unset HIPSOXR_FFT_NO_WAVE; export HIPSOXR_DEBUG_WAVE_MIN=1; [ "$v" != "$name" ] && export HIPSOXR_FFT_NO_WAVE=1
  echo "== $v"; HIPSOXR_LIBRARY=$V/$name/libhipsoxr.so timeout 300 python tools/wave_check.py ${CLIPS:-128} ${SECS:-10} ${RATES:-48000 44100} 2>&1 | tail -1
done; done
} > gpurun_out/r6_ab.txt 2>&1
cat gpurun_out/r6_ab.txt
