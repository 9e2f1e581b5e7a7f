#!/usr/bin/env python
"""Host-side cost of one hipsoxr_run_device call (configs[1] job, frequency-domain engine): calls issued back to back
without waiting; while the queue is not full the loop runs at the host's pace, then at the GPU's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch
from soxr_amd import device as dev
plan = dev.Plan(48000, 44100, "VHQ")
x = torch.randn(48000 * 60, device="cuda") * 0.25
y = dev.resample_tensor(plan, x)
job = dev.PreparedJob(plan, x, y)
for _ in range(20): job.launch()
torch.cuda.synchronize()
for n in (8, 32, 128, 1024, 8192):
    t0 = time.perf_counter()
    for _ in range(n): job.launch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n:5d} launches: issue {1e6 * (t1 - t0) / n:6.2f} us per call, until done {1e6 * (t2 - t0) / n:6.2f} us per call")
