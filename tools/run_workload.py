#!/usr/bin/env python
"""Launch ONE bench workload N times and nothing else (profiling target: the kernel of interest is the only hipsoxr
kernel in the trace).  run_workload.py <batch|clip|c2|f64|i32> [launches] [kernel-id]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch
from soxr_amd import device as dev
what = sys.argv[1] if len(sys.argv) > 1 else "batch"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = torch.Generator(device="cuda"); g.manual_seed(7)
if what == "batch":
    plan = dev.Plan(48000, 44100, "VHQ")
    x = torch.randn((128, 480000, 1), device="cuda", generator=g) * 0.25
elif what == "clip":
    plan = dev.Plan(48000, 44100, "VHQ")
    x = torch.randn(2880000, device="cuda", generator=g) * 0.25
elif what == "c2":
    plan = dev.Plan(44100, 16000, "VHQ")
    x = torch.randn((2646000, 8), device="cuda", generator=g) * 0.25
elif what == "f64":
    plan = dev.Plan(48000, 44100, "VHQ")
    x = torch.randn(2880000, device="cuda", dtype=torch.float64, generator=g) * 0.25
elif what == "i32":
    plan = dev.Plan(48000, 44100, "VHQ")
    x = (torch.randn(2880000, device="cuda", dtype=torch.float64, generator=g) * 0.25 * 2 ** 30).to(torch.int32)
else:
    raise SystemExit("unknown workload " + what)
if os.environ.get("ZERO_INPUT"):  # DVFS probe: the same launch on zero-filled input (less switching activity: higher clock if power-bound)
    x = torch.zeros_like(x)
y = dev.resample_tensor(plan, x, kernel=kernel)
job = dev.PreparedJob(plan, x, y, kernel=kernel)
ROT = int(os.environ.get("ROTATE", "0"))   # ROTATE=n: cycle through n distinct input / output buffers (nothing of a launch's input is cache-resident from the launch before)
if ROT > 1:
    xs = [x] + [torch.randn(x.shape, device="cuda", dtype=torch.float32).to(x.dtype) * 0.25 for _ in range(ROT - 1)]
    ys = [y] + [torch.empty_like(y) for _ in range(ROT - 1)]
    jobs = [dev.PreparedJob(plan, a, b, kernel=kernel) for a, b in zip(xs, ys)]
    class _R:
        i = 0
        def launch(self):
            jobs[self.i % ROT].launch(); self.i += 1
    job = _R()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    job.launch()
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    job.launch()
e1.record(); torch.cuda.synchronize()
print("%s: %.2f us per launch over %d launches" % (what, e0.elapsed_time(e1) * 1e3 / n, n))
