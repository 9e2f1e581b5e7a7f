#!/bin/bash
# round 6: the one-wave-per-pair kernel against k_fft_pair2 on one box (batch shard), then the FFT suites with every
# eligible job forced onto the wave kernel (debug build, HIPSOXR_DEBUG_WAVE_MIN=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DBG=$PWD/python-soxr_amd/_variants/dbg/libhipsoxr.so
{
timeout 300 python tools/wave_check.py 128 10
HIPSOXR_LIBRARY=$DBG HIPSOXR_FFT_NO_WAVE=1 timeout 300 python tools/wave_check.py 128 10
timeout 300 python tools/wave_check.py 128 10
HIPSOXR_LIBRARY=$DBG HIPSOXR_FFT_NO_WAVE=1 timeout 300 python tools/wave_check.py 128 10
timeout 300 python tools/wave_check.py 128 10 44100 48000
HIPSOXR_LIBRARY=$DBG HIPSOXR_FFT_NO_WAVE=1 timeout 300 python tools/wave_check.py 128 10 44100 48000
HIPSOXR_LIBRARY=$DBG HIPSOXR_DEBUG_WAVE_MIN=1 timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r6_wave.txt 2>&1
tail -40 gpurun_out/r6_wave.txt
