#!/usr/bin/env python
"""Randomised check of the one-wave-per-pair kernel (csrc/fftwave.hip) against the float64 oracle: 48k <-> 44.1k, HQ / VHQ,
random lengths (around block and pair boundaries too), clip counts, planar channel counts, input and output columns at
random 4-byte phases, ragged batches.  Run it on the debug-switch build with every eligible job forced onto the kernel:
    HIPSOXR_LIBRARY=python-soxr_amd/_variants/dbg/libhipsoxr.so HIPSOXR_DEBUG_WAVE_MIN=1 python tests/fuzz/fuzz_fft_wave.py [cases] [seed]
(without the switch the product's own rule applies and small jobs run on k_fft_pair2: still a valid check, of another kernel).
Bar: 1e-6 relative RMS, exact shapes, nothing written outside a column."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from soxr_amd import device as dev
from soxr_amd import dist as sdist
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FFT = 5
fails = 0


def bad(y, ref):
    if y.shape != ref.shape:
        return True
    if not ref.size:
        return False
    return not np.sqrt(np.mean((y - ref) ** 2)) <= 1e-6 * max(np.sqrt(np.mean(ref ** 2)), 1e-3)


for case in range(n_cases):
    fi, fo = r.choice([(48000, 44100), (44100, 48000)])
    q = r.choice(["VHQ", "HQ"])
    hop = 3234 if fo == 44100 else 3520
    n_out = r.choice([r.randint(1, 200), hop * r.randint(1, 6) + r.randint(-2, 2), r.randint(200, 120000)])
    n = max(1, int(n_out * fi / fo))
    rng = np.random.default_rng(1000 + case)
    plan = dev.Plan(fi, fo, q)
    kind = r.choice(["planar", "phase", "ragged"])
    if kind == "planar":
        clips, ch = r.choice([1, 2, 5]), r.choice([1, 2, 3])
        x = (rng.standard_normal((clips, ch, n)) * 0.25).astype(np.float32)
        y = dev.resample_tensor(plan, torch.from_numpy(x).cuda().permute(0, 2, 1), kernel=FFT).cpu().numpy()
        ok = not any(bad(y[c, :, k], oracle.resample(x[c, k].astype(np.float64), fi, fo, q, mode="ref")) for c in range(clips) for k in range(ch))
    elif kind == "phase":
        pi, po = r.randint(0, 3), r.randint(0, 3)
        big = torch.from_numpy((rng.standard_normal(n + 8) * 0.25).astype(np.float32)).cuda()
        xin = big[pi:pi + n]
        no = plan.out_len(n)
        ybuf = torch.zeros(no + 8, device="cuda")
        yv = ybuf[po:po + no]
        dev.PreparedJob(plan, xin.view(1, -1, 1), yv.view(1, -1, 1), kernel=FFT).launch()
        torch.cuda.synchronize()
        ok = not bad(yv.cpu().numpy(), oracle.resample(xin.cpu().numpy().astype(np.float64), fi, fo, q, mode="ref"))
        ok = ok and float(ybuf[:po].abs().sum()) == 0.0 and float(ybuf[po + no:].abs().sum()) == 0.0
    else:
        lens = [r.choice([0, r.randint(1, 50), r.randint(50, 40000)]) for _ in range(r.randint(2, 6))] + [n]
        clips = [torch.from_numpy((rng.standard_normal(m) * 0.25).astype(np.float32)).cuda() for m in lens]
        job = sdist.RaggedJob(plan, clips, kernel=FFT)
        job.launch()
        torch.cuda.synchronize()
        outs = job.outputs()
        ok = not any(bad(outs[i].cpu().numpy().reshape(-1), oracle.resample(clips[i].cpu().numpy().astype(np.float64), fi, fo, q, mode="ref")) for i in range(len(lens)))
    if not ok:
        fails += 1
        print(f"FAIL case {case}: {fi}->{fo} {q} {kind} n={n}")
print(f"fft-wave fuzz: {fails} failures in {n_cases} cases")
sys.exit(1 if fails else 0)
