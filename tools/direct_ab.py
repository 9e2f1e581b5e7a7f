#!/usr/bin/env python
"""One-shot soxr.resample (host arrays) against the size above which results stop being written by the kernel straight
into pinned host memory (HIPSOXR_DEBUG_DIRECT_MAX, set by the caller's environment): best of 7, us."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import soxr_amd as soxr
rng = np.random.default_rng(0)
def best(f, n=7):
    f(); b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b
for a, b in ((48000, 44100), (44100, 48000), (16000, 48000)):
    for dt in ("float32", "int16"):
        for ch in (1, 2, 8):
            row = []
            for n in (2000, 5000, 10000, 20000, 50000, 100000):
                x = rng.standard_normal((n, ch)) * 0.25
                x = (x * 20000).astype(np.int16) if dt == "int16" else x.astype(np.float32)
                if ch == 1: x = x[:, 0].copy()
                row.append("%7.1f" % (best(lambda: soxr.resample(x, a, b, "VHQ")) * 1e6))
            print(f"{a}->{b} {dt:8s} ch={ch}: frames 2k/5k/10k/20k/50k/100k:", " ".join(row), flush=True)
