#!/usr/bin/env python
"""Per-call cost of hipsoxr_stream_process itself (ctypes, no Python surface) for small chunks."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
from soxr_amd import _native as n
for flags, chunk in ((0, 441), (0, 4410), (n.DEFER, 441), (n.RESIDENT, 441), (n.RESIDENT, 100), (n.RESIDENT, 4410)):
    h = C.c_void_p()
    n.check(n.lib.hipsoxr_stream_create(44100.0, 16000.0, 1, n.I16, n.VHQ, flags, C.byref(h)))
    x = (np.random.default_rng(0).standard_normal(chunk) * 5000).astype(np.int16)
    y = np.empty(chunk, np.int16)
    done = C.c_size_t()
    for _ in range(50):
        n.lib.hipsoxr_stream_process(h, x.ctypes.data, chunk, y.ctypes.data, chunk, C.byref(done))
    t0 = time.perf_counter(); calls = 2000
    for _ in range(calls):
        n.lib.hipsoxr_stream_process(h, x.ctypes.data, chunk, y.ctypes.data, chunk, C.byref(done))
    dt = (time.perf_counter() - t0) / calls
    print(f"flags {flags:3d} chunk {chunk}: {dt * 1e6:.1f} us per hipsoxr_stream_process call ({chunk / dt / 1e6:.2f} Msamples/s)")
    n.lib.hipsoxr_stream_delete(h)
