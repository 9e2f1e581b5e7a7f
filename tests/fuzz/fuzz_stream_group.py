#!/usr/bin/env python
"""Randomised check of many-streams calls (soxr_amd.device.TensorStreamGroup over hipsoxr_streams_process_device) against the
same streams called singly on HOST arrays (ResampleStream, which the other fuzzers pin to the oracle): random rate pairs
(standard, integer, float), recipes, dtypes, 1-3 channels, 2-40 handles de-phased by random prefixes, distinct dither
seeds, random chunk sizes per round from 1 to 6000 frames (small ones take the shared launch, large ones the per-handle
path inside the same call; the ring moves inside launches along the way), a member used alone now and then — the same
frames for every handle in every call, bit for bit.
`python tests/fuzz/fuzz_stream_group.py [cases] [seed]`"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "python-soxr_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
import soxr_amd as soxr
from soxr_amd import device as dev

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for case in range(n_cases):
    in_rate = r.choice([44100, 48000, 96000, 16000, r.randint(8000, 96000), r.uniform(8000, 96000)])
    out_rate = r.choice([44100, 48000, 16000, 22050, r.randint(8000, 96000), in_rate / r.uniform(0.5, 4.0)])
    q = r.choice(["VHQ", "HQ", "MQ", "LQ", "QQ"])
    dtype = r.choice([np.float32, np.float64, np.int16, np.int32])
    ch = r.choice([1, 1, 2, 3])
    n = r.randint(2, 40)
    tdt = torch.from_numpy(np.zeros(1, dtype)).dtype
    seeds = [r.randint(0, 2 ** 31) for _ in range(n)]
    grp = dev.TensorStreamGroup(n, in_rate, out_rate, ch, dtype=tdt, quality=q, dither_seeds=seeds)
    ref = [soxr.ResampleStream(in_rate, out_rate, ch, dtype=dtype, quality=q, dither_seed=seeds[i]) for i in range(n)]
    rng = np.random.default_rng(case)

    def sig(shape):
        x = rng.standard_normal(shape)
        return (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)

    ok = True
    for i in range(n):  # de-phase
        pre = sig((r.randint(0, 700), ch) if ch > 1 else r.randint(0, 700))
        a = grp.streams[i].resample_chunk(torch.from_numpy(pre).cuda()).cpu().numpy()
        ok = ok and np.array_equal(a, ref[i].resample_chunk(pre))
    for rnd in range(r.randint(2, 25)):
        if not ok:
            break
        frames = r.choice([1, r.randint(2, 500), 441, 441, r.randint(500, 2500), r.randint(2500, 6000)])
        x = sig((n, frames, ch) if ch > 1 else (n, frames))
        y, counts = grp.resample_chunks(torch.from_numpy(x).cuda())
        y = y.cpu().numpy()
        for i in range(n):
            w = ref[i].resample_chunk(x[i])
            if counts[i] != len(w) or not np.array_equal(y[i, :counts[i]], w):
                ok = False
                print(f"FAIL case {case} round {rnd} stream {i}/{n}: {in_rate!r}->{out_rate!r} {q} {np.dtype(dtype).name} ch={ch} frames={frames} got {counts[i]} want {len(w)}")
                break
        if ok and r.random() < 0.2:  # a member alone between group calls
            i = r.randrange(n)
            x1 = sig((r.randint(1, 300), ch) if ch > 1 else r.randint(1, 300))
            ok = np.array_equal(grp.streams[i].resample_chunk(torch.from_numpy(x1).cuda()).cpu().numpy(), ref[i].resample_chunk(x1))
    if ok:  # flush a few members
        for i in r.sample(range(n), min(n, 3)):
            z = sig((0, ch) if ch > 1 else 0)
            ok = ok and np.array_equal(grp.streams[i].resample_chunk(torch.from_numpy(z).cuda(), last=True).cpu().numpy(), ref[i].resample_chunk(z, last=True))
    fails += not ok
print(f"stream-group fuzz: {fails} failures in {n_cases} cases")
sys.exit(1 if fails else 0)
