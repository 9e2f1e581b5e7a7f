#!/usr/bin/env python
"""Latency of the host-pointer surface on SHORT inputs: soxr.resample (a stream per call, from the pool) against a
reused ResampleStream (clear + one flushing call), and the pieces of a call (create / process / delete)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np, soxr_amd as soxr
from soxr_amd import _native as nat
rng = np.random.default_rng(0)
def best(f, n=50, reps=5):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n): f()
        b = min(b, (time.perf_counter() - t0) / n)
    return b * 1e6
for n in (480, 4800, 48000, 480000):
    x = (rng.standard_normal(n) * 0.25).astype(np.float32)
    for _ in range(5): soxr.resample(x, 48000, 44100, quality="VHQ")
    rs = soxr.ResampleStream(48000, 44100, 1, quality="VHQ")
    def reuse():
        rs.clear(); rs.resample_chunk(x, last=True)
    h = C.c_void_p()
    def create_delete():
        nat.check(nat.lib.hipsoxr_stream_create(48000.0, 44100.0, 1, nat.FLOAT32_I, nat.VHQ, 0, C.byref(h)))
        nat.lib.hipsoxr_stream_delete(h)
    print(f"{n:7d} frames: soxr.resample {best(lambda: soxr.resample(x, 48000, 44100, quality='VHQ')):7.1f} us   reused stream (clear + chunk) {best(reuse):7.1f} us   "
          f"stream create + delete {best(create_delete):7.1f} us")
