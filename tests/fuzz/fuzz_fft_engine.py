#!/usr/bin/env python
"""Randomised check of the frequency-domain engine (device API, AUTO/FFT) against the float64
oracle: random ratios from the schedule table and outside it, lengths, channel counts, layouts
(interleaved / planar / strided views, also starting at an odd channel of a wider tensor / batched), float32 and
float64.  Bar: 1e-6 relative RMS (float64 VHQ: 5e-9), exact shapes.
`python tests/fuzz/fuzz_fft_engine.py [cases] [seed]`"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from soxr_amd import device as dev
from oracle import oracle
import _provider  # noqa: F401  (port mode on the product's bank: arithmetic-order check)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
STD = [8000, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000]
fails = 0
for case in range(n_cases):
    i, o = r.choice(STD), r.choice(STD)
    if i == o or o / i > 12 or i / o > 24:
        continue
    q = r.choice(["VHQ", "HQ"])
    ch = r.choice([1, 2, 2, 3, 4, 8])
    clips = r.choice([1, 1, 2, 5])
    n = r.choice([1, 50, r.randint(100, 3000), r.randint(3000, 60000), r.randint(60000, 250000)])
    if o > 2 * i:
        n = min(n, 40000)
    rng = np.random.default_rng(case)
    dt = r.choice([np.float32, np.float32, np.float64])
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
    kernel = r.choice([dev.KERNEL_AUTO, dev.KERNEL_FFT])
    try:
        y = dev.resample_tensor(plan, xt, kernel=kernel).cpu().numpy()
    except RuntimeError as e:
        if "FFT engine unavailable" in str(e) or "FFT engine needs" in str(e):
            continue
        print(f"FAIL case {case}: {i}->{o} {q} clips={clips} n={n} ch={ch} {layout}: {e}"); fails += 1; continue
    ok = True
    for c in range(clips):
        ref = oracle.resample(x[c].astype(np.float64), i, o, q, mode="ref")
        if y[c].shape != ref.shape:
            ok = False; break
        if ref.size:
            err = np.sqrt(np.mean((y[c] - ref) ** 2)); rms = max(np.sqrt(np.mean(ref ** 2)), 1e-3)
            if not err <= (5e-9 if (dt is np.float64 and q == "VHQ" and kernel == dev.KERNEL_FFT) else 1e-6) * rms:
                ok = False; break
    if not ok:
        fails += 1
        print(f"FAIL case {case}: {i}->{o} {q} {dt.__name__} clips={clips} n={n} ch={ch} {layout} kernel={kernel}")
print(f"fft-engine fuzz: {fails} failures in {n_cases} cases")
sys.exit(1 if fails else 0)
