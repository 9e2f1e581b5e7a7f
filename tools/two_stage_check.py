#!/usr/bin/env python
"""Two-stage form (twostage.hip) of interpolated-phase plans: AUTO against the exact engine — relative RMS, head and tail,
and the launch times of both.  tools/two_stage_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np, torch
from soxr_amd import device as dev

def timeit(plan, x, kernel, n=30):
    y = dev.resample_tensor(plan, x, kernel=kernel)
    job = dev.PreparedJob(plan, x, y, kernel=kernel)
    for _ in range(3): job.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): job.launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, y

g = torch.Generator(device="cuda"); g.manual_seed(3)
cases = [(48000, 44101, "VHQ", 2880000, 2, torch.float32), (44101, 48000, "VHQ", 2646060, 2, torch.float32), (48000, 44101, "HQ", 2880000, 1, torch.float32),
         (44100, 16001, "VHQ", 2646000, 2, torch.float32), (16001, 44100, "VHQ", 960060, 1, torch.float32), (48000, 44101, "VHQ", 480000, 1, torch.float64),
         (44101, 48000, "VHQ", 441010, 2, torch.float64), (48000.5, 32000.25, "VHQ", 1000000, 1, torch.float32), (8000, 44101, "VHQ", 500000, 1, torch.float32)]
for a, b, q, frames, ch, dt in cases:
    plan = dev.Plan(a, b, q)
    x = (torch.randn((frames, ch), device="cuda", generator=g, dtype=torch.float32) * 0.25).to(dt)
    if ch == 1: x = x[:, 0].contiguous()
    t_exact, ye = timeit(plan, x, dev.KERNEL_EXACT, 10)
    t_auto, ya = timeit(plan, x, dev.KERNEL_AUTO, 30)
    d = (ya.double() - ye.double()).cpu().numpy(); e = ye.double().cpu().numpy()
    rel = np.sqrt((d ** 2).mean() / (e ** 2).mean())
    w = 400
    print("%9.2f -> %9.2f %s %-7s %dch %8d frames  phases %d taps %d: exact %7.1f us  auto %7.1f us  rel rms %.2e  head %.2e tail %.2e max %.2e" % (
        a, b, q, str(dt).split('.')[-1], ch, frames, plan.phases, plan.taps, t_exact, t_auto, rel,
        np.sqrt((d[:w] ** 2).mean()) / 0.25, np.sqrt((d[-w:] ** 2).mean()) / 0.25, np.abs(d).max()))
