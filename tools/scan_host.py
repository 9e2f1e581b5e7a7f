#!/usr/bin/env python
"""Host-surface scan (numpy in, numpy out; PCIe-inclusive): soxr.resample over sizes x channels x dtypes x ratios and
ResampleStream over chunk sizes — looks for cliffs, not for headline numbers."""
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
print("== soxr.resample (one-shot), best of 7, us")
for a, b in ((48000, 44100), (44100, 16000), (44100, 48000), (16000, 48000)):
    for dt in ("float32", "int16"):
        for ch in (1, 2):
            row = []
            for n in (1000, 10000, 100000, 1000000):
                x = rng.standard_normal((n, ch)) * 0.25
                x = (x * 20000).astype(np.int16) if dt == "int16" else x.astype(np.float32)
                if ch == 1: x = x[:, 0].copy()
                row.append("%8.1f" % (best(lambda: soxr.resample(x, a, b, "VHQ")) * 1e6))
            print(f"{a}->{b} {dt:8s} ch={ch}: frames 1k/10k/100k/1M:", " ".join(row), flush=True)
print("== ResampleStream per call, us (2000 / 400 / 100 calls)")
for a, b in ((48000, 44100), (44100, 16000), (44100, 48000)):
    for dt in ("float32", "int16"):
        for ch in (1, 2):
            row = []
            for chunk, calls in ((128, 2000), (480, 2000), (1024, 1000), (4800, 400), (48000, 100)):
                x = rng.standard_normal((chunk, ch)) * 0.25
                x = (x * 20000).astype(np.int16) if dt == "int16" else x.astype(np.float32)
                if ch == 1: x = x[:, 0].copy()
                rs = soxr.ResampleStream(a, b, ch, dtype=dt, quality="VHQ")
                for _ in range(20): rs.resample_chunk(x)
                t0 = time.perf_counter()
                for _ in range(calls): rs.resample_chunk(x)
                row.append("%7.1f" % ((time.perf_counter() - t0) / calls * 1e6))
            print(f"{a}->{b} {dt:8s} ch={ch}: chunk 128/480/1024/4800/48000:", " ".join(row), flush=True)
