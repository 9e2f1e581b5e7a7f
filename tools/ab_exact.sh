#!/bin/bash
# exact engine (kernel 6) on the configs[1] shape, float32 and int16: ab_exact.sh [ENV=...]
for e in "${@:-A=1}"; do for dt in f32 i16; do
echo -n "[$e] $dt: "; env $e DTYPE=$dt python tools/time_config.py 48000 44100 VHQ 2880000 1 1 6 2>&1 | tail -1 | cut -c1-60
done; done
