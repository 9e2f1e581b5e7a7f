#!/usr/bin/env python
"""per-call time of small-chunk streams, interleaved vs split layout (44100 -> 16000 int16 VHQ stereo, 441-frame chunks):
plain / resident / deferred — tools/split_stream_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from soxr_amd import _native as n
from test_gpu_split_streams import _feed
rng = np.random.default_rng(3)
x = (rng.standard_normal((44100 * 4, 2)) * 5000).astype(np.int16)
for name, flags in (("plain", 0), ("resident", n.RESIDENT), ("deferred", n.DEFER)):
    t = {s: min(_feed(n, x, 44100, 16000, n.I16, s, flags, 441, 6)[2] for _ in range(3)) for s in (False, True)}
    print("%-9s interleaved %.2f us per call   split %.2f us per call" % (name, t[False] * 1e6, t[True] * 1e6))
