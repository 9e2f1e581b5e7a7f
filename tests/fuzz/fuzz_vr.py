#!/usr/bin/env python
"""Randomised check of variable-rate streams against the oracle driven by the integer clock
restatement (tests/vr_sim.py): random largest ratio, recipe, dtype, chunk sizes, ratio changes with
random slew lengths (including changes during a slew and zero-length chunks), one to three interleaved channels, and
— one chunk in five — chunks of up to 120 000 frames (launches of thousands of outputs: k_interp_wave).  Bit-exact, chunk
by chunk and channel by channel.  `python tests/fuzz/fuzz_vr.py [cases] [seed]`"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "python-soxr_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import soxr_amd as soxr
from oracle import oracle
import _provider  # noqa: F401  (port mode on the product's bank: arithmetic-order check)
from vr_sim import VrSim

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
only = int(os.environ.get('ONLY', '-1'))
for case in range(n_cases):
    in_rate = r.choice([44100, 48000, 96000, r.uniform(8000, 96000)])
    max_io = r.choice([1.0, 1.5, 2.0, 3.0, r.uniform(0.6, 6.0)])
    out_rate = in_rate / max_io
    q = r.choice(["VHQ", "HQ", "MQ", "LQ", "QQ"])
    dtype = r.choice([np.float32, np.float64, np.int16, np.int32])
    ch = r.choice([1, 1, 2, 3])
    live = only < 0 or only == case
    if live:
        rs = soxr.ResampleStream(in_rate, out_rate, ch, dtype=dtype, quality=q, vr=True)
        sims = [VrSim(oracle, in_rate, out_rate, q, dtype) for _ in range(ch)]
    rng = np.random.default_rng(case)
    n_chunks = r.randint(2, 9)
    ok = True
    hist = []
    for c in range(n_chunks):
        n = r.choice([0, 1, r.randint(2, 400), r.randint(400, 6000), r.randint(400, 6000), r.randint(6000, 120000)])
        x = rng.standard_normal((n, ch) if ch > 1 else n)
        x = (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)
        last = c == n_chunks - 1
        if live and ok:
            y = rs.resample_chunk(x, last=last)
            if ch == 1:
                w = sims[0].feed(x, last=last)
            else:
                cols = [sims[k].feed(x[:, k], last=last, channel=k) for k in range(ch)]
                w = np.stack(cols, axis=1) if len(cols[0]) else np.zeros((0, ch), dtype)
            if only >= 0:
                print(f"  chunk {c}: n={n} got {len(y)} want {len(w)} equal={np.array_equal(y, w)} delay={rs.delay():.3f} k_done={sims[0].k_done}")
        if live and ok and (len(y) != len(w) or not np.array_equal(y, w)):
            ok = False
            print(f"FAIL case {case} chunk {c}: in_rate={in_rate!r} io0={max_io!r} {q} {np.dtype(dtype).name} ch={ch} n={n} got {len(y)} want {len(w)} history={hist}")
        if r.random() < 0.6 and not last:
            io = r.uniform(0.3, 1.0) * max_io
            slew = r.choice([0, 0, 1, r.randint(2, 3000)])
            hist.append((c, n, io, slew))
            if live and ok:
                if only >= 0:
                    print(f"  set_io_ratio({io!r}, slew={slew})")
                rs.set_io_ratio(io, 1.0, slew)
                for sm in sims:
                    sm.set_io_ratio(io / 1.0, slew)
    fails += not ok
print(f"vr fuzz: {fails} failures in {n_cases} cases")
sys.exit(1 if fails else 0)
