#!/usr/bin/env python
"""Debug aid: per-call time of 441-frame ResampleStream calls (44100 -> 16000 int16 VHQ) after a history of other streams
in the same process (does a stream still turn resident by itself?).  tools/vr_stream_time.py [legs...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import soxr_amd as soxr
rng = np.random.default_rng(5)
x = (rng.standard_normal(44100 * 20) * 5000).astype(np.int16)
KW = {"cr": dict(), "res": dict(resident=True), "def": dict(deferred=True), "vr": dict(vr=True)}
def leg(name, chunk):
    rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ", **KW[name])
    rs.resample_chunk(x[:chunk]); rs.clear()
    n = 0; t0 = time.perf_counter()
    for a in range(0, len(x), chunk):
        rs.resample_chunk(x[a:a + chunk], last=(a + chunk >= len(x))); n += 1
    print(f"{name:4s} chunk {chunk:6d}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call", flush=True)
for spec in sys.argv[1:]:
    name, chunk = spec.split(":")
    leg(name, int(chunk))
