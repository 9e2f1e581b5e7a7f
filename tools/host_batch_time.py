#!/usr/bin/env python
"""bench.py's host_batch leg alone, over block sizes: tools/host_batch_time.py [n_clips]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import bench
from soxr_amd import dist as sdist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
print(bench.host_batch(n))
rng = np.random.default_rng(11)
lens = rng.integers(5 * 48000, 15 * 48000 + 1, size=n)
pool = (rng.standard_normal(15 * 48000 + n) * 0.25).astype(np.float32)
clips = [pool[i:i + int(m)].copy() for i, m in enumerate(lens)]
for bb in (16 << 20, 32 << 20, 64 << 20, 128 << 20):
    sdist.resample_batch(clips[:64], 48000, 44100, "VHQ", devices=[0], block_bytes=bb)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], block_bytes=bb); best = min(best, time.perf_counter() - t0)
    print("block %4d MB: %.1f ms" % (bb >> 20, best * 1e3))
# the two host copies alone (no GPU): what the CPU side costs
t0 = time.perf_counter()
tmp = [c.copy() for c in clips]
print("plain copy of the corpus, 1 thread: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
