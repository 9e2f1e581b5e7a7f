"""Helper of tests/test_gpu_switches.py: run a fixed set of jobs under the process's HIPSOXR_* environment and print
one JSON line of digests (exact engines: SHA-256; frequency-domain engine: the float64 sum of squares of the
difference to the exact engine's result on the same input, relative)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
import torch  # noqa: E402
import soxr_amd as soxr  # noqa: E402
from soxr_amd import device as dev  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-9))


rng = np.random.default_rng(99)
out = {}
# canonical-order engine: host surface, streams (plain / resident / deferred), device jobs of several shapes
x1 = (rng.standard_normal(30000) * 0.25).astype(np.float32)
out["host_f32"] = sha(soxr.resample(x1, 48000, 44100, quality="VHQ"))
xi = (rng.standard_normal((9000, 2)) * 5000).astype(np.int16)
out["host_i16"] = sha(soxr.resample(xi, 44100, 16000, quality="HQ"))
out["host_interp"] = sha(soxr.resample(x1[:8000], 48000, 44101.5, quality="HQ"))
out["host_interp_2ch"] = sha(soxr.resample(xi, 44100, 16000.5, quality="VHQ"))       # 3265 outputs x 2 channels
vs = soxr.ResampleStream(44100, 16000, 1, dtype="float32", quality="VHQ", vr=True)        # chunks of 7256 and ~10 000 outputs
v0 = vs.resample_chunk(x1[:20000])
vs.set_io_ratio(44100, 22050, 700)
out["stream_vr"] = sha(np.concatenate([v0, vs.resample_chunk(x1[20000:], last=True)]))
rs = soxr.ResampleStream(44100, 16000, 2, dtype="int16", quality="VHQ")                    # 20 000-frame chunks: k_gather_wave
xl = (rng.standard_normal((50000, 2)) * 5000).astype(np.int16)
out["stream_20000"] = sha(np.concatenate([rs.resample_chunk(xl[a:a + 20000], last=(a + 20000 >= len(xl))) for a in range(0, len(xl), 20000)]))
xg = torch.from_numpy((rng.standard_normal((9000, 3)) * 0.25).astype(np.float32)).cuda()  # forced gather path, >= 4096 outputs
out["dev_gather"] = sha(dev.resample_tensor(dev.Plan(48000, 44100, "HQ"), xg, kernel=dev.KERNEL_GATHER).cpu().numpy())
for name, kw in (("stream", {}), ("stream_resident", {"resident": True}), ("stream_deferred", {"deferred": True})):
    rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ", **kw)
    parts = [rs.resample_chunk(xi[a:a + 441, 0].copy(), last=(a + 441 >= len(xi))) for a in range(0, len(xi), 441)]
    out[name] = sha(np.concatenate(parts))
xd = torch.from_numpy((rng.standard_normal((4, 60000, 1)) * 0.25).astype(np.float32)).cuda()
plan = dev.Plan(48000, 44100, "VHQ")
exact = dev.resample_tensor(plan, xd, kernel=dev.KERNEL_EXACT).cpu().numpy()
out["dev_exact"] = sha(exact)
x64 = torch.from_numpy(rng.standard_normal(50000) * 0.25).cuda()
out["dev_exact_f64"] = sha(dev.resample_tensor(plan, x64, kernel=dev.KERNEL_EXACT).cpu().numpy())
x8 = torch.from_numpy((rng.standard_normal((40000, 8)) * 0.25).astype(np.float32)).cuda()
plan2 = dev.Plan(44100, 16000, "VHQ")
exact8 = dev.resample_tensor(plan2, x8, kernel=dev.KERNEL_EXACT).cpu().numpy()
out["dev_exact_8ch"] = sha(exact8)
# frequency-domain engine (AUTO): relative distance to the exact result
out["fft_batch"] = rel(dev.resample_tensor(plan, xd).cpu().numpy(), exact)
out["fft_8ch"] = rel(dev.resample_tensor(plan2, x8).cpu().numpy(), exact8)
xl = torch.from_numpy((rng.standard_normal((40, 200000, 1)) * 0.25).astype(np.float32)).cuda()   # 40 x 21 pairs: a "large" job
yl = dev.resample_tensor(plan, xl)
out["fft_large_sha"] = sha(yl.cpu().numpy())
out["fft_large"] = rel(yl[3, :, 0].cpu().numpy(), dev.resample_tensor(plan, xl[3, :, 0].contiguous(), kernel=dev.KERNEL_EXACT).cpu().numpy())
# interpolated-phase plans at the tile kernel's sizes (round 5: two outputs per lane — channel pairs / a column's two halves):
# canonical order, so bit for bit whatever serves them; and the two-stage form (AUTO) within 1e-6 of it
plan3 = dev.Plan(48000, 44101, "VHQ")
xt = torch.from_numpy((rng.standard_normal((300000, 3)) * 0.25).astype(np.float32)).cuda()
e1 = dev.resample_tensor(plan3, xt[:, 0].contiguous(), kernel=dev.KERNEL_EXACT)
e2 = dev.resample_tensor(plan3, xt[:, :2].contiguous(), kernel=dev.KERNEL_EXACT)
e3 = dev.resample_tensor(plan3, (xt * 20000).to(torch.int16), kernel=dev.KERNEL_EXACT, dither=True)
out["dev_interp_tile"] = sha(e1.cpu().numpy()) + sha(e2.cpu().numpy()) + sha(e3.cpu().numpy())
out["two_stage_mono"] = rel(dev.resample_tensor(plan3, xt[:, 0].contiguous()).cpu().numpy(), e1.cpu().numpy())
out["two_stage_2ch"] = rel(dev.resample_tensor(plan3, xt[:, :2].contiguous()).cpu().numpy(), e2.cpu().numpy())
torch.cuda.synchronize()
print("SWITCH_PROBE " + json.dumps(out))
