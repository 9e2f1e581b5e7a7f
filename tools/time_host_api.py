#!/usr/bin/env python
"""Host-pointer surface timings (PCIe-inclusive): soxr.resample on numpy arrays and the
configs[4] streaming pattern (int16, 44.1k -> 16k, chunked)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import soxr_amd as soxr

rng = np.random.default_rng(0)
x = (rng.standard_normal(48000 * 60) * 0.25).astype(np.float32)
soxr.resample(x[:48000], 48000, 44100, "VHQ")
for name, arr, a, b, q in [("configs[1] 60 s mono f32 VHQ 48k->44.1k", x, 48000, 44100, "VHQ"),
                           ("configs[0] 10 s mono f32 HQ 48k->44.1k", x[:480000], 48000, 44100, "HQ")]:
    soxr.resample(arr, a, b, q)
    t0 = time.perf_counter(); n = 10
    for _ in range(n):
        soxr.resample(arr, a, b, q)
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:8.3f} ms per call  {len(arr) / dt / 1e6:9.1f} Msamples/s (host to host)")
x8 = (rng.standard_normal((44100 * 60, 8)) * 0.25).astype(np.float32)
soxr.resample(x8, 44100, 16000, "VHQ")
t0 = time.perf_counter(); soxr.resample(x8, 44100, 16000, "VHQ"); dt = time.perf_counter() - t0
print(f"configs[2] 60 s x 8 ch f32 VHQ 44.1k->16k: {dt * 1e3:8.3f} ms per call  {x8.size / dt / 1e6:9.1f} Msamples/s")
xi = (rng.standard_normal((44100 * 60, 1)) * 5000).astype(np.int16)
for chunk in (441, 4410, 96000):
    rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ")
    t0 = time.perf_counter(); calls = 0
    for i in range(0, len(xi), chunk):
        rs.resample_chunk(xi[i:i + chunk], last=(i + chunk >= len(xi))); calls += 1
    dt = time.perf_counter() - t0
    print(f"configs[4] stream int16 mono, chunk {chunk:6d}: {dt * 1e3:8.1f} ms total, {dt / calls * 1e6:8.1f} us per call, {len(xi) / dt / 1e6:8.2f} Msamples/s")
