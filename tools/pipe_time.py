#!/usr/bin/env python
"""soxr.resample on long host clips (10-120 s mono float32): ms per call and GB/s moved, both directions summed.
(Round 4 used it to A/B a pipelined one-shot against the plain one: profiles/NOTES_r04.md §6.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import soxr_amd as soxr
rng = np.random.default_rng(0)
x = (rng.standard_normal(48000 * 120) * 0.25).astype(np.float32)
for sec in (10, 20, 30, 60, 120):
    a = x[:48000 * sec]
    soxr.resample(a, 48000, 44100, "VHQ")
    best = 1e9
    for _ in range(10):
        t0 = time.perf_counter(); soxr.resample(a, 48000, 44100, "VHQ"); best = min(best, time.perf_counter() - t0)
    print(f"{sec:4d} s mono f32 VHQ 48k->44.1k: {best * 1e3:7.3f} ms  ({(a.nbytes + a.nbytes * 147 // 160) / best / 1e9:5.1f} GB/s both ways)", flush=True)
