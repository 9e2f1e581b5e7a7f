#!/usr/bin/env python
"""Times one device-resident float32 job: tools/time_config.py in_rate out_rate quality frames channels [clips] [kernel...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch
from soxr_amd import device as dev

in_rate, out_rate, q = float(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
frames, ch = int(sys.argv[4]), int(sys.argv[5])
clips = int(sys.argv[6]) if len(sys.argv) > 6 else 1
kernels = [int(k) for k in sys.argv[7:]] or [0, 6]
plan = dev.Plan(in_rate, out_rate, q)
DTYPE = {"f32": torch.float32, "f64": torch.float64, "i16": torch.int16, "i32": torch.int32}[os.environ.get("DTYPE", "f32")]
x = torch.randn((clips, frames, ch), device="cuda") * 0.25
x = (x * 20000).to(DTYPE) if DTYPE in (torch.int16, torch.int32) else x.to(DTYPE)
print(f"plan L={plan.L} M={plan.M} T={plan.taps} phases={plan.phases}")
for k in kernels:
    try:
        y = dev.resample_tensor(plan, x, kernel=k)
    except RuntimeError as e:
        print(f"kernel {k}: {e}")
        continue
    job = dev.PreparedJob(plan, x, y, kernel=k)   # one C call per launch: Python overhead stays below the kernel time
    for _ in range(3):
        job.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        job.launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    nbytes = x.element_size() * (x.numel() + y.numel())
    print(f"kernel {k}: {us:9.1f} us  {x.numel() / us:9.1f} Msamples/s in  {nbytes / us / 1e3:7.1f} GB/s algorithmic ({nbytes / us / 1e3 / 8000:.3f} of 8 TB/s)")
