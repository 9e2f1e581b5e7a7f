#!/usr/bin/env python
"""one two-stage job launched N times (profiling target): tools/two_stage_prof.py in out quality frames ch [n] [exact]
(`exact`: the same job on the canonical-order engine — k_interp_tile)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch
from soxr_amd import device as dev
a, b, q, frames, ch = float(sys.argv[1]), float(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
plan = dev.Plan(a, b, q)
x = torch.randn((frames, ch), device="cuda") * 0.25
if ch == 1: x = x[:, 0].contiguous()
kern = dev.KERNEL_EXACT if len(sys.argv) > 7 and sys.argv[7] == "exact" else dev.KERNEL_AUTO
y = dev.resample_tensor(plan, x, kernel=kern)
job = dev.PreparedJob(plan, x, y, kernel=kern)
for _ in range(n): job.launch()
torch.cuda.synchronize()
