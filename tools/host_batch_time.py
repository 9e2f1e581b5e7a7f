#!/usr/bin/env python
"""bench.py's host_batch leg, then the same corpus over copy-thread counts and block sizes, results in pinned
buffers and in pageable arrays: tools/host_batch_time.py [n_clips]"""
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


def best_of(k=3, **kw):
    outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], **kw)
    best = 1e9
    for _ in range(k):
        outs = None
        t0 = time.perf_counter(); outs = sdist.resample_batch(clips, 48000, 44100, "VHQ", devices=[0], **kw); best = min(best, time.perf_counter() - t0)
    return best * 1e3


for pinned in (True, False):
    for th in (4, 6, 8, 12, 14):
        os.environ["SOXR_AMD_COPY_THREADS"] = str(th); sdist._PIPES.clear()
        print("pinned results %d, %2d copy threads: %.1f ms" % (pinned, th, best_of(pinned_results=pinned)), flush=True)
os.environ.pop("SOXR_AMD_COPY_THREADS"); sdist._PIPES.clear()
for bb in (16 << 20, 32 << 20, 64 << 20, 128 << 20):
    print("pinned results, block %4d MB: %.1f ms" % (bb >> 20, best_of(block_bytes=bb, pinned_results=True)), flush=True)
t0 = time.perf_counter()
tmp = [c.copy() for c in clips]
print("plain copy of the corpus, 1 thread: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
