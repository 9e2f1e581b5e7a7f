#!/bin/bash
# k_gather_wave against the period-tile kernels on exact-bank jobs that under-fill the chip (debug-switch build):
# device jobs of a few thousand to a few hundred thousand outputs, and the 96 000-frame stream calls.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HIPSOXR_LIBRARY=python-soxr_amd/_variants/dbg/libhipsoxr.so
for spec in "44100 16000 VHQ 96000 1 i16" "44100 16000 VHQ 300000 1 i16" "48000 44100 VHQ 48000 1 f32" "48000 44100 VHQ 480000 1 f32" "48000 44100 VHQ 200000 2 f32" "48000 44100 HQ 480000 2 f32" "44100 48000 VHQ 441000 1 f64" "44100 16000 VHQ 1000000 1 i16"; do
  set -- $spec
  for sw in "HIPSOXR_NO_GATHER_WAVE=1" "HIPSOXR_DEBUG_GW_TAPS=100000"; do
    printf "%-34s %-28s " "$spec" "$sw"
    env $sw DTYPE=$6 python tools/time_config.py $1 $2 $3 $4 $5 1 6 2>&1 | grep "kernel 6" | cut -c1-40
  done
done
python tools/vr_stream_time.py cr:4410 cr:20000 cr:96000
HIPSOXR_NO_GATHER_WAVE=1 python tools/vr_stream_time.py cr:4410 cr:20000 cr:96000
