#!/usr/bin/env python
"""The one-wave-per-pair FFT kernel (csrc/fftwave.hip) on the batch shard: results against the exact engine, launch time
on rotating buffer sets.  tools/wave_check.py [clips] [seconds] [in_rate out_rate]
Run it twice for an A/B on one box: plain, and with HIPSOXR_LIBRARY=<dbg build> HIPSOXR_FFT_NO_WAVE=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import numpy as np
import torch
from soxr_amd import device as dev

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 128
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
in_rate = float(sys.argv[3]) if len(sys.argv) > 3 else 48000.
out_rate = float(sys.argv[4]) if len(sys.argv) > 4 else 44100.
frames = int(in_rate * secs)
plan = dev.Plan(in_rate, out_rate, "VHQ")
torch.manual_seed(1)
NSETS = 3
KERNEL = int(os.environ.get('KERNEL', '5'))   # 5 = frequency-domain engine, 8 = the same on float64 arithmetic, 6 = exact engine
xs = [torch.randn((clips, frames, 1), device="cuda") * 0.25 for _ in range(NSETS)]
ys = [dev.resample_tensor(plan, x, kernel=KERNEL) for x in xs]
# correctness: a few clips against the exact engine (bit-exact with the oracle: tests/test_gpu_parity.py)
for c in sorted({0, 1, clips // 2, clips - 1}):
    ex = dev.resample_tensor(plan, xs[0][c:c + 1], kernel=6).double()
    d = ys[0][c:c + 1].double() - ex
    rel = float(d.pow(2).mean().sqrt() / ex.pow(2).mean().sqrt())
    print(f"clip {c}: rel rms vs exact {rel:.3e}  max {float(d.abs().max()):.3e}  head {float(d[0, :4000].abs().max()):.2e} tail {float(d[0, -4000:].abs().max()):.2e}")
    assert rel < 1e-6 or os.environ.get('NOCHECK'), rel
jobs = [dev.PreparedJob(plan, x, y, kernel=KERNEL) for x, y in zip(xs, ys)]
for j in jobs:
    j.launch()
torch.cuda.synchronize()
res = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for i in range(n):
        jobs[i % NSETS].launch()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 1e3 / n)
nbytes = 4 * (xs[0].numel() + ys[0].numel())
us = float(np.median(res))
print(f"{'NO_WAVE' if os.environ.get('HIPSOXR_FFT_NO_WAVE', '') else 'wave   '} clips {clips} x {secs} s: {us:8.2f} us per launch (runs {', '.join('%.1f' % r for r in res)})"
      f"  {nbytes / 1e6:.1f} MB  {nbytes / us / 1e6 / 8:.3f} of 8 TB/s")
