#!/usr/bin/env python
"""Debug aid: run one batch launch of k_tile_mfma_p with HIPSOXR_DEBUG_TRACE and summarise the
per-wave s_memtime stamps: 0 start, 1 staged, 2 barrier, 3.. after each unit of the first slab,
then end of first slab, 15 end."""
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
y = dev.resample_tensor(plan, x, kernel=4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
y = dev.resample_tensor(plan, x, kernel=4)
e1.record()
torch.cuda.synchronize()
t = np.fromfile("/tmp/hipsoxr_trace.bin", dtype=np.uint64).reshape(-1, 4, 16).astype(np.int64)
start, end = t[:, :, 0], t[:, :, 15]
span = end.max() - start.min()
print("workgroups", t.shape[0], "span ticks", span, "event ms (incl. trace dump)", e0.elapsed_time(e1))
life = end.max(axis=1) - start.min(axis=1)
print("workgroup lifetime median", np.median(life), "p10", np.percentile(life, 10), "p90", np.percentile(life, 90))
print("mean resident workgroups per CU (sum lifetimes / span / 256): %.2f" % (life.sum() / span / 256))
valid = t[:, :, 1:9]
seg = np.diff(t[:, :, 0:9], axis=2)
print("median segments [stage, barrier, unit0..4, slab end]:", [int(np.median(seg[:, :, i])) for i in range(8)])
# WG start times histogram
st = np.sort(start.min(axis=1) - start.min())
print("WG start deciles (ticks):", [int(np.percentile(st, p)) for p in range(0, 101, 10)])
