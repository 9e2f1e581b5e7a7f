#!/usr/bin/env python
"""Power and clock under ONE workload: launches back to back for a few seconds while a thread samples
`rocm-smi --showpower --showclocks` (GPU box).  tools/power_probe.py <workload> [seconds]
Environment switches (debug build through HIPSOXR_LIBRARY / tools/with_variant.sh) select the kernel; ZERO_INPUT=1
feeds zeros (less switching activity)."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch
from soxr_amd import device as dev
what = sys.argv[1] if len(sys.argv) > 1 else "batch"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
g = torch.Generator(device="cuda"); g.manual_seed(7)
plan = dev.Plan(48000, 44100, "VHQ")
kernel = int(os.environ.get("KERNEL", "0"))
if what == "batch":
    x = torch.randn((128, 480000, 1), device="cuda", generator=g) * 0.25
elif what == "clip":
    x = torch.randn(2880000, device="cuda", generator=g) * 0.25
else:
    raise SystemExit("workload?")
if os.environ.get("ZERO_INPUT"):
    x = torch.zeros_like(x)
y = dev.resample_tensor(plan, x, kernel=kernel)
job = dev.PreparedJob(plan, x, y, kernel=kernel)
torch.cuda.synchronize()
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:
            out = ""
        pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([0-9.]+)", out)
        sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        mclk = re.findall(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
        samples.append((pw[:1], sclk[:8], mclk[:1]))
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(500):
        job.launch()
    n += 500
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
us = e0.elapsed_time(e1) * 1e3 / n
mid = samples[len(samples) // 3:] or samples
pws = [float(s[0][0]) for s in mid if s[0]]
scl = [sum(map(int, s[1])) / len(s[1]) for s in mid if s[1]]
print("%s: %.2f us per launch (%d launches); power W: %s  sclk MHz: %s  (%d samples)" % (
    what, us, n, ("%.0f" % (sum(pws) / len(pws))) if pws else "?", ("%.0f" % (sum(scl) / len(scl))) if scl else "?", len(samples)))
if pws:
    print("   energy per launch: %.1f mJ" % (sum(pws) / len(pws) * us * 1e-3))
