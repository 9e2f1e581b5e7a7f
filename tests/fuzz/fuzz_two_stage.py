#!/usr/bin/env python
"""Randomised check of the two-stage form of arbitrary ratios (csrc/twostage.hip, device API under AUTO / FFT) against the
canonical-order engine — itself bit-identical to the oracle's float64 direct form on the same bank (fuzz_device_exact.py):
random integer / float rate pairs (some a hair off 1, 1/2, 2, 4), HQ / VHQ, float32 / float64, 1-8 channels, 1-5 clips,
lengths from below the form's threshold to a few hundred thousand frames, interleaved / planar / strided / offset views.
Bars: exact shapes; relative RMS <= 1e-6 (float64 VHQ: 3e-9) over every column; the first and last 128 outputs at the same
absolute scale (x 8).  `python tests/fuzz/fuzz_two_stage.py [cases] [seed]`"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from soxr_amd import device as dev

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = took = 0
for case in range(n_cases):
    kind = r.choice(["int", "float", "near"])
    if kind == "int":
        i, o = r.randint(8000, 96000), r.randint(8000, 96000)
    elif kind == "float":
        i, o = r.uniform(8000, 96000), r.uniform(8000, 96000)
    else:
        i = r.choice([16000, 22050, 44100, 48000])
        o = i * r.choice([1, .5, 2, 4, .25, 3, 1 / 3]) * (1 + r.choice([-1, 1]) * r.choice([1e-6, 1e-5, 3e-4, 2e-3]))
    if not 0.2 < o / i < 12:
        continue
    q = r.choice(["VHQ", "HQ"])
    ch = r.choice([1, 2, 2, 3, 4, 6, 8])
    clips = r.choice([1, 1, 2, 5])
    n = r.choice([5000, r.randint(8000, 20000), r.randint(20000, 120000), r.randint(120000, 400000)])
    if o > 3 * i or clips * ch > 12:
        n = min(n, 60000)
    dt = r.choice([np.float32, np.float32, np.float64])
    rng = np.random.default_rng(1000 + case)
    x = (rng.standard_normal((clips, n, ch)) * 0.25).astype(dt)
    layout = r.choice(["interleaved", "planar", "strided", "offset"])
    xt = torch.from_numpy(x).cuda()
    if layout == "planar":
        xt = xt.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    elif layout == "strided" and ch > 1:
        xt = torch.from_numpy(np.concatenate([x, x], axis=2)).cuda()[:, :, :ch]
    elif layout == "offset":
        pad = r.choice([1, 2, 3])
        wide = np.concatenate([x[:, :, :1].repeat(pad, axis=2) * 0 + 7, x, x[:, :, :1] * 0 - 7], axis=2)
        xt = torch.from_numpy(wide).cuda()[:, :, pad:pad + ch]
    plan = dev.Plan(i, o, q)
    kernel = r.choice([dev.KERNEL_AUTO, dev.KERNEL_AUTO, dev.KERNEL_FFT])
    tag = f"case {case}: {i}->{o} {q} {dt.__name__} clips={clips} n={n} ch={ch} {layout} kernel={kernel} phases={plan.phases}"
    try:
        y = dev.resample_tensor(plan, xt, kernel=kernel).double().cpu().numpy()
    except RuntimeError as e:
        if "FFT engine unavailable" in str(e) or "FFT engine needs" in str(e):
            continue
        print("FAIL", tag, e); fails += 1; continue
    ye = dev.resample_tensor(plan, xt, kernel=dev.KERNEL_EXACT).double().cpu().numpy()
    if y.shape != ye.shape:
        print("FAIL shape", tag, y.shape, ye.shape); fails += 1; continue
    if not ye.size or np.array_equal(y, ye):
        continue                      # (not a job the two-stage form takes: the exact engine served it)
    took += 1
    d = y - ye
    scale = np.maximum(np.sqrt(np.mean(ye ** 2, axis=1)), 1e-3)          # per (clip, channel)
    rel = (np.sqrt(np.mean(d ** 2, axis=1)) / scale).max()
    ends = (np.maximum(np.abs(d[:, :128]).max(axis=1), np.abs(d[:, -128:]).max(axis=1)) / scale).max()
    # (HQ's interpolated plan has 32 intervals: its own table is a 2.5e-7 approximation of the prototype the FFT stage samples exactly)
    bar = 3e-9 if (dt is np.float64 and q == "VHQ") else 1e-6
    if not (rel <= bar and ends <= 8 * bar):
        print("FAIL", tag, "rel %.3g ends %.3g" % (rel, ends)); fails += 1
print(f"two-stage fuzz: {fails} failures in {n_cases} cases ({took} took the two-stage form)")
sys.exit(1 if fails or not took else 0)
