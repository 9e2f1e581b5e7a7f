#!/usr/bin/env python
"""Debug aid: one batch launch of k_fft_pair2 built with -DFFT2_TRACE and HIPSOXR_DEBUG_TRACE set; summarises
the per-wave s_memtime stamps.  Stamp index: 0 start | forward: 1 pass-1 done (input arrived + butterfly +
LDS stores), 2 barrier, 3 pass-2 done, 4 barrier, 5 pass-3 done | 6 barrier | inverse: 7, 8, 9, 10, 11 likewise
(11 = staging written) | 12 barrier | 15 end (run stored).
    HIPSOXR_EXTRA_FLAGS=-DFFT2_TRACE bash python-soxr_amd/build.sh && python tools/trace_pair2.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
os.environ["HIPSOXR_DEBUG_TRACE"] = "/tmp/hipsoxr_trace.bin"
import torch
from soxr_amd import device as dev
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
plan = dev.Plan(48000, 44100, "VHQ")
x = torch.randn((clips, frames, 1), device="cuda") * 0.25
for _ in range(3):
    y = dev.resample_tensor(plan, x)
    torch.cuda.synchronize()
t = np.fromfile("/tmp/hipsoxr_trace.bin", dtype=np.uint64).reshape(-1, 6, 16).astype(np.int64)
start, end = t[:, :, 0], t[:, :, 15]
span = end.max() - start.min()
# s_memtime counts shader cycles; the columns below are cycles / 100 (so 21.0 = 2100 cycles = 1 us at 2.1 GHz).
# Counters of different XCDs are not synchronised: spans across workgroups mean nothing, lifetimes do.
print("workgroups", t.shape[0])
life = end.max(axis=1) - start.min(axis=1)
print("workgroup lifetime (cycles/100) median %.2f  p10 %.2f  p90 %.2f" % tuple(np.percentile(life, [50, 10, 90]) / 100))
names = ["F1(load+bfly+st)", "bar", "F2", "bar", "F3", "bar", "I1", "bar", "I2", "bar", "I3+stage", "bar", "store-out"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15]
mid = t[t.shape[0] // 4: 3 * t.shape[0] // 4]          # steady state: the middle half of the launch
for w in range(6):
    seg = np.diff(mid[:, w, idx], axis=1)
    print("wave %d median cycles/100:" % w, " ".join("%s=%.2f" % (n, np.median(seg[:, i]) / 100) for i, n in enumerate(names)))

