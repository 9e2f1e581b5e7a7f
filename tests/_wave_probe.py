"""Helper of tests/test_gpu_fft_wave.py: jobs of the frequency-domain engine that the one-wave-per-pair kernel
(csrc/fftwave.hip) serves, under the process's HIPSOXR_* environment; prints one JSON line of relative RMS errors
against the oracle's float64 direct form on its own bank (mode "ref") and of digests."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch  # noqa: E402
from soxr_amd import device as dev  # noqa: E402
from soxr_amd import dist as sdist  # noqa: E402
from oracle import oracle as o  # noqa: E402

FFT = 5


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-3)) if a.shape == b.shape else 9.0


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


rng = np.random.default_rng(2026)
out = {}
for name, (fi, fo) in {"down": (48000, 44100), "up": (44100, 48000)}.items():
    plan = dev.Plan(fi, fo, "VHQ")
    # lengths around the kept run of a block (3234 / 3520 outputs) and of a pair: single blocks, odd block counts, edges
    for n in (1, 7, 100, 3519, 3520, 3521, 7040, 7041, 3234 * 3 + 5, 50001):
        x = (rng.standard_normal(n) * 0.25).astype(np.float32)
        y = dev.resample_tensor(plan, torch.from_numpy(x).cuda(), kernel=FFT).cpu().numpy()
        out[f"{name}_len_{n}"] = rel(y, o.resample(x, fi, fo, "VHQ", mode="ref"))
    # a batch of planar stereo clips (unit frame stride), HQ
    planq = dev.Plan(fi, fo, "HQ")
    xb = (rng.standard_normal((3, 2, 40000)) * 0.25).astype(np.float32)
    xt = torch.from_numpy(xb).cuda().permute(0, 2, 1)                        # [clip, frame, channel], channel-major memory
    yb = dev.resample_tensor(planq, xt, kernel=FFT).cpu().numpy()
    out[f"{name}_planar_hq"] = max(rel(yb[c, :, ch], o.resample(xb[c, ch], fi, fo, "HQ", mode="ref")) for c in range(3) for ch in range(2))
    # columns that start at every 4-byte phase of a 16-byte granule, input and output (8-byte loads at 4-byte alignment; the
    # staged run's first granule element by element)
    big = torch.from_numpy((rng.standard_normal(4 + 30011) * 0.25).astype(np.float32)).cuda()
    worst = 0.0
    for off in range(4):
        xin = big[off:off + 30007]
        nout = plan.out_len(30007)
        ybuf = torch.zeros(nout + 8, device="cuda")
        yv = ybuf[(off + 1) % 4:(off + 1) % 4 + nout]
        dev.PreparedJob(plan, xin.view(1, -1, 1), yv.view(1, -1, 1), kernel=FFT).launch()
        torch.cuda.synchronize()
        worst = max(worst, rel(yv.cpu().numpy(), o.resample(xin.cpu().numpy(), fi, fo, "VHQ", mode="ref")))
        assert float(ybuf[:(off + 1) % 4].abs().sum()) == 0.0 and float(ybuf[(off + 1) % 4 + nout:].abs().sum()) == 0.0   # nothing outside the column
    out[f"{name}_phases"] = worst
    # ragged batch: every clip where it lies, lengths from nothing to several pairs
    lens = [0, 5, 3000, 7041, 20000, 12345, 33333]
    clips = [torch.from_numpy((rng.standard_normal(n) * 0.25).astype(np.float32)).cuda() for n in lens]
    job = sdist.RaggedJob(plan, clips, kernel=FFT)
    job.launch()
    torch.cuda.synchronize()
    outs = job.outputs()
    out[f"{name}_ragged"] = max([rel(outs[i].cpu().numpy().reshape(-1), o.resample(clips[i].cpu().numpy(), fi, fo, "VHQ", mode="ref")) for i in range(len(lens)) if lens[i]] +
                                [0.0 if outs[0].numel() == 0 else 9.0])
    # determinism and a digest (the same job twice)
    xd = torch.from_numpy((rng.standard_normal((4, 60000, 1)) * 0.25).astype(np.float32)).cuda()
    y1 = dev.resample_tensor(plan, xd, kernel=FFT).cpu().numpy()
    y2 = dev.resample_tensor(plan, xd, kernel=FFT).cpu().numpy()
    out[f"{name}_deterministic"] = bool(np.array_equal(y1, y2))
    out[f"{name}_sha"] = sha(y1)
    out[f"{name}_vs_exact"] = rel(y1, dev.resample_tensor(plan, xd, kernel=6).cpu().numpy())
print("WAVE_PROBE " + json.dumps(out))
