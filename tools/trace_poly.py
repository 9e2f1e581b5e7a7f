#!/usr/bin/env python
"""Debug aid: one two-stage job on a build with -DPOLY_TRACE -DHIPSOXR_DEBUG_SWITCHES; per-wave cycle sums of k_poly by phase.
    HIPSOXR_VARIANT=ptrace HIPSOXR_EXTRA_FLAGS="-DPOLY_TRACE -DHIPSOXR_DEBUG_SWITCHES" bash python-soxr_amd/build.sh   (here)
    tools/with_variant.sh ptrace python tools/trace_poly.py [in out frames channels]                                   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
os.environ["HIPSOXR_DEBUG_TRACE"] = "/tmp/hipsoxr_ptrace.bin"
import torch
from soxr_amd import device as dev
a, b = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (48000., 44101.)
frames, ch = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2880000, 2)
plan = dev.Plan(a, b, "VHQ")
x = torch.randn((frames, ch), device="cuda") * 0.25
for _ in range(3):
    y = dev.resample_tensor(plan, x); torch.cuda.synchronize()
t = np.fromfile("/tmp/hipsoxr_ptrace.bin", dtype=np.uint64).reshape(-1, 4, 8).astype(np.int64)
print("workgroups", t.shape[0])
life = t[:, :, 6] - t[:, :, 5]
print("wave lifetime cycles: median %d p10 %d p90 %d" % tuple(np.percentile(life, [50, 10, 90])))
names = ["stage span", "barrier", "compute", "barrier", "store"]
tot = t[:, :, :5].sum(axis=2)
for i, n in enumerate(names):
    print("%-11s median %8d cycles  (%.1f %% of the phases' sum)" % (n, np.median(t[:, :, i]), 100. * t[:, :, i].sum() / tot.sum()))
print("unaccounted (table load, prologue): median %d" % np.median(life - tot))
