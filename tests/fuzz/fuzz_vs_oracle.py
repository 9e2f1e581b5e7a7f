#!/usr/bin/env python
"""Randomised differential test: soxr_amd (GPU) against the CPU oracle, bit for bit, over random
rates (standard, random integer, random float), dtypes, recipes, lengths, channel counts, layouts,
one-shot and chunked.  `python tests/fuzz/fuzz_vs_oracle.py [cases] [seed]`"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd")); sys.path.insert(0, ROOT)
import numpy as np
import soxr_amd as soxr
from oracle import oracle
import _provider  # noqa: F401  (port mode on the product's bank: arithmetic-order check)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
STD = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000]
fails = 0
for case in range(n_cases):
    kind = r.choice(["std", "std", "int", "float"])
    if kind == "std":
        i, o = r.choice(STD), r.choice(STD)
    elif kind == "int":
        i, o = r.randint(8000, 96000), r.randint(8000, 96000)
    else:
        i, o = r.uniform(8000, 96000), r.uniform(8000, 96000)
    q = r.choice(["VHQ", "HQ", "MQ", "LQ", "QQ"])
    dtype = r.choice([np.float32, np.float64, np.int16, np.int32])
    ch = r.choice([1, 1, 2, 3, 8])
    n = r.choice([0, 1, 2, 7, 100, 1000, r.randint(2, 30000), r.randint(2, 30000)])
    if o / i > 8 and n > 5000:
        n = 5000
    rng = np.random.default_rng(case)
    x = rng.standard_normal((n, ch))
    x = (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)
    mode = r.choice(["resample", "fortran", "stream", "mono1d"])
    try:
        want = oracle.resample(x, i, o, q, mode="port")
        if mode == "resample":
            got = soxr.resample(x, i, o, quality=q)
        elif mode == "fortran":
            got = soxr.resample(np.asfortranarray(x), i, o, quality=q)
        elif mode == "mono1d":
            got = soxr.resample(x[:, 0], i, o, quality=q); want = want[:, 0]
        else:
            rs = soxr.ResampleStream(i, o, ch, dtype=dtype, quality=q)
            chunk = r.choice([1, 17, 441, 5000])
            parts = [np.empty((0, ch), dtype)]
            if n == 0:
                parts.append(rs.resample_chunk(x, last=True))
            for a in range(0, n, chunk):
                parts.append(rs.resample_chunk(x[a:a + chunk], last=(a + chunk >= n)))
            got = np.concatenate(parts)
        ok = got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want)
    except Exception as e:  # noqa: BLE001
        ok = False
        got = e
    if not ok:
        fails += 1
        print(f"FAIL case {case}: {i}->{o} {q} {np.dtype(dtype).name} ch={ch} n={n} {mode}: {got if isinstance(got, Exception) else (got.shape, want.shape)}")
print(f"{n_cases - fails}/{n_cases} cases bit-identical to the oracle")
sys.exit(1 if fails else 0)
