#!/usr/bin/env python
"""Randomised check of the exact engine through the device API (resample_tensor, KERNEL_EXACT and
the explicit kernels): random dtypes, ratios (standard / arbitrary), batches, channel counts,
layouts — bit-identical to the oracle's canonical-order port.
`python tests/fuzz/fuzz_device_exact.py [cases] [seed]`"""
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
STD = [8000, 11025, 12000, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000]
TORCH = {np.float32: torch.float32, np.float64: torch.float64, np.int16: torch.int16, np.int32: torch.int32}
fails = 0
for case in range(n_cases):
    if r.random() < 0.7:
        i, o = r.choice(STD), r.choice(STD)
    else:
        i, o = r.uniform(8000, 96000), r.randint(8000, 96000)
    if o / i > 6:
        continue
    q = r.choice(["VHQ", "HQ", "MQ", "LQ", "QQ"])
    dtype = r.choice([np.float32, np.float64, np.int16, np.int32])
    ch, clips = r.choice([1, 2, 3, 5]), r.choice([1, 1, 3])
    n = r.choice([1, 37, r.randint(100, 5000), r.randint(5000, 40000), r.randint(40000, 120000), r.randint(40000, 120000),
                  r.randint(300000, 1200000)])  # (the last: few-slab launches of the tile kernels, row-tile splits)
    if n > 300000:
        ch, clips = r.choice([1, 2]), 1
    rng = np.random.default_rng(case)
    x = rng.standard_normal((clips, n, ch))
    x = (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)
    xt = torch.from_numpy(x).cuda()
    layout = r.choice(["interleaved", "planar"])
    if layout == "planar":
        xt = xt.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    plan = dev.Plan(i, o, q)
    kernel = r.choice([dev.KERNEL_EXACT, dev.KERNEL_EXACT, dev.KERNEL_GATHER, dev.KERNEL_TILE, dev.KERNEL_TILE_VALU])
    try:
        y = dev.resample_tensor(plan, xt, kernel=kernel).cpu().numpy()
    except RuntimeError as e:
        if "tile kernel unavailable" in str(e):
            continue
        print(f"FAIL case {case}: {e}"); fails += 1; continue
    ok = all(np.array_equal(y[c], oracle.resample(x[c], i, o, q, mode="port", dither=False)) for c in range(clips))
    if not ok:
        fails += 1
        print(f"FAIL case {case}: {i}->{o} {q} {np.dtype(dtype).name} clips={clips} n={n} ch={ch} {layout} kernel={kernel}")
print(f"device exact-engine fuzz: {fails} failures in {n_cases} cases")
sys.exit(1 if fails else 0)
