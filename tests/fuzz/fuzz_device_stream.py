#!/usr/bin/env python
"""Randomised check of device-chunk streams (soxr_amd.device.TensorStream over hipsoxr_stream_process_device) against the
host-array stream (ResampleStream, which the other fuzzers and tests pin to the oracle): random rate pairs (standard,
integer, float), recipes, dtypes, 1-3 channels, constant and variable rate (random ratio changes and slews), random chunk
sizes from 0 to 30 000 frames, sometimes on a side torch stream — the same frames in the same calls, bit for bit.
`python tests/fuzz/fuzz_device_stream.py [cases] [seed]`"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "python-soxr_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
import soxr_amd as soxr
from soxr_amd import device as dev

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
side = torch.cuda.Stream()
for case in range(n_cases):
    in_rate = r.choice([44100, 48000, 96000, 16000, r.randint(8000, 96000), r.uniform(8000, 96000)])
    out_rate = r.choice([44100, 48000, 16000, 22050, r.randint(8000, 96000), in_rate / r.uniform(0.5, 4.0)])
    q = r.choice(["VHQ", "HQ", "MQ", "LQ", "QQ"])
    dtype = r.choice([np.float32, np.float64, np.int16, np.int32])
    ch = r.choice([1, 1, 2, 3])
    vr = r.random() < 0.4
    use_side = r.random() < 0.3
    rs = soxr.ResampleStream(in_rate, out_rate, ch, dtype=dtype, quality=q, vr=vr)
    ts = dev.TensorStream(in_rate, out_rate, ch, dtype=torch.from_numpy(np.zeros(1, dtype)).dtype, quality=q, vr=vr)
    rng = np.random.default_rng(case)
    n_chunks = r.randint(1, 8)
    ok = True
    for c in range(n_chunks):
        n = r.choice([0, 1, r.randint(2, 500), r.randint(500, 6000), r.randint(6000, 30000)])
        x = rng.standard_normal((n, ch) if ch > 1 else n)
        x = (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)
        last = c == n_chunks - 1
        w = rs.resample_chunk(x, last=last)
        if use_side:
            with torch.cuda.stream(side):
                y = ts.resample_chunk(torch.from_numpy(x).cuda(), last=last)
            side.synchronize()
        else:
            y = ts.resample_chunk(torch.from_numpy(x).cuda(), last=last)
        y = y.cpu().numpy()
        if y.shape != w.shape or not np.array_equal(y, w):
            ok = False
            print(f"FAIL case {case} chunk {c}: {in_rate!r}->{out_rate!r} {q} {np.dtype(dtype).name} ch={ch} vr={vr} n={n} got {y.shape} want {w.shape}")
            break
        if vr and not last and r.random() < 0.6:
            io = r.uniform(0.3, 1.0) * in_rate / out_rate
            slew = r.choice([0, 0, 1, r.randint(2, 3000)])
            rs.set_io_ratio(io, 1.0, slew)
            ts.set_io_ratio(io, 1.0, slew)
    fails += not ok
print(f"device-stream fuzz: {fails} failures in {n_cases} cases")
sys.exit(1 if fails else 0)
