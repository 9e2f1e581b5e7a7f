#!/bin/bash
# round 6: k_tile_mfma_p launch forms on the 60 s mono clip (HIPSOXR_DEBUG_TILE_FORM 1..4: slab 64 whole / 64 split / 32 whole / 32 split)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
DBG=$PWD/python-soxr_amd/_variants/dbg/libhipsoxr.so
{
for rep in 1 2; do
echo "== default"; HIPSOXR_LIBRARY=$DBG timeout 300 python tools/time_config.py 48000 44100 VHQ 2880000 1 1 6 2>&1 | grep "kernel 6"
for f in 1 2 3 4; do echo "== form $f"; HIPSOXR_DEBUG_TILE_FORM=$f HIPSOXR_LIBRARY=$DBG timeout 300 python tools/time_config.py 48000 44100 VHQ 2880000 1 1 6 2>&1 | grep "kernel 6"; done
for z in 2 3 4 5; do echo "== split $z"; HIPSOXR_DEBUG_SPLIT=$z HIPSOXR_LIBRARY=$DBG timeout 300 python tools/time_config.py 48000 44100 VHQ 2880000 1 1 6 2>&1 | grep "kernel 6"; done
done
} > gpurun_out/r6_forms.txt 2>&1
cat gpurun_out/r6_forms.txt
