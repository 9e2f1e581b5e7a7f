#!/usr/bin/env python
"""k_fft_pair2<.., P = 2> against P = 1 and the exact engine: relative RMS over odd sizes (GPU box, debug-switch build).
tools/x2_check.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "python-soxr_amd", "_variants", "ntsweep", "libhipsoxr.so")  # HIPSOXR_VARIANT=ntsweep HIPSOXR_EXTRA_FLAGS="-DFFT_NT_SWEEP -DFFT_EXPERIMENT_X2 -DHIPSOXR_DEBUG_SWITCHES" bash python-soxr_amd/build.sh
CHILD = r'''
import sys, os, json, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "python-soxr_amd"))
from soxr_amd import device as dev
out = {}
rng = np.random.default_rng(3)
for (a, b) in ((48000, 44100), (44100, 48000)):
    plan = dev.Plan(a, b, "VHQ")
    for frames, clips in ((480000, 3), (123457, 5), (9000, 2), (2880001, 1), (40000, 40)):
        x = torch.from_numpy((rng.standard_normal((clips, frames, 1)) * 0.25).astype(np.float32)).cuda()
        y = dev.resample_tensor(plan, x).cpu().numpy()
        e = dev.resample_tensor(plan, x, kernel=dev.KERNEL_EXACT).cpu().numpy()
        d = (y.astype(np.float64) - e)
        out["%d>%d %dx%d" % (a, b, clips, frames)] = [float(np.sqrt((d ** 2).mean() / (e.astype(np.float64) ** 2).mean())), float(np.abs(d).max()), list(y.shape)]
print("X2CHECK " + json.dumps(out))
'''
for name, env in (("P=1", {"HIPSOXR_FFT_X2": "0"}), ("P=2", {"HIPSOXR_FFT_X2": "1"}), ("P=2 small", {"HIPSOXR_FFT_X2": "1", "HIPSOXR_FFT_SMALL_ONLY": "1", "HIPSOXR_FFT_NO_TINY": "1"})):
    e = dict(os.environ); e.update(env); e["HIPSOXR_LIBRARY"] = DBG
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=e, capture_output=True, text=True)
    if r.returncode:
        print(name, "FAILED", r.stderr[-1500:]); continue
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("X2CHECK ")][-1][8:])
    print(name)
    for k, v in d.items():
        print("   %-26s rel rms %.3e  max abs %.3e  %s" % (k, v[0], v[1], v[2]))
