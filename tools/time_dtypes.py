#!/usr/bin/env python
"""soxr.resample host-API time per dtype (10 s and 60 s mono, 48k -> 44.1k, HQ and VHQ)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import soxr_amd as soxr
rng = np.random.default_rng(0)
base = rng.standard_normal(48000 * 60)
for dtype in (np.float32, np.float64, np.int16, np.int32):
    x = (base * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (base * 0.25).astype(dtype)
    for q in ("HQ", "VHQ"):
        for secs in (10, 60):
            arr = x[:48000 * secs]
            soxr.resample(arr, 48000, 44100, q)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); soxr.resample(arr, 48000, 44100, q); best = min(best, time.perf_counter() - t0)
            print(f"{np.dtype(dtype).name:8s} {q:3s} {secs:2d} s: {best * 1e3:7.3f} ms  {len(arr) / best / 1e6:8.1f} Msamples/s")
