#!/usr/bin/env python
"""Debug aid: one batch launch of k_fft_wave (csrc/fftwave.hip) built with -DFFT2_TRACE and HIPSOXR_DEBUG_TRACE set;
summarises the per-wave s_memtime stamps.  Stamp index: 0 start | 1 loads + forward pass 0 | 2 exchange | 3 twiddles +
forward pass 1 | 4 spectrum exchange | 5 inverse pass 0 | 6 exchange | 7 twiddles + inverse pass 1 | 8 run a stored |
9 run b stored | 13 HW_ID | 14 XCC_ID | 15 end.
    HIPSOXR_VARIANT=trace HIPSOXR_EXTRA_FLAGS="-DFFT2_TRACE -DHIPSOXR_DEBUG_SWITCHES" bash python-soxr_amd/build.sh   (here)
    tools/with_variant.sh trace python tools/trace_wave.py                                  (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
os.environ["HIPSOXR_DEBUG_TRACE"] = "/tmp/hipsoxr_trace.bin"
import torch
from soxr_amd import device as dev
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
plan = dev.Plan(48000, 44100, "VHQ")
x = torch.randn((clips, frames, 1), device="cuda") * 0.25
for _ in range(3):
    y = dev.resample_tensor(plan, x, kernel=5)
    torch.cuda.synchronize()
t = np.fromfile("/tmp/hipsoxr_trace.bin", dtype=np.uint64).reshape(-1, 16).astype(np.int64)
t = t[t[:, 15] != 0]
print("waves", t.shape[0])
life = t[:, 15] - t[:, 0]
print("wave lifetime (cycles/100) median %.1f  p10 %.1f  p90 %.1f" % tuple(np.percentile(life, [50, 10, 90]) / 100))
names = ["load+F0", "xchg", "tw+F1", "spec", "I0", "xchg", "tw+I1", "run a", "run b"]
mid = t[t.shape[0] // 4: 3 * t.shape[0] // 4]
seg = np.diff(mid[:, :10], axis=1)
print("median cycles/100:", " ".join("%s=%.1f" % (n, np.median(seg[:, i]) / 100) for i, n in enumerate(names)))
print("p90    cycles/100:", " ".join("%s=%.1f" % (n, np.percentile(seg[:, i], 90) / 100) for i, n in enumerate(names)))
hw, xcc = t[:, 13], t[:, 14] & 0xF
simd = (hw >> 4) & 0x3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cus = np.unique(key)
cnt = np.bincount(np.searchsorted(cus, key))
print("distinct CUs seen: %d; waves per CU: min %d median %d max %d" % (len(cus), cnt.min(), np.median(cnt), cnt.max()))
conc, spans, gaps = [], [], []
for c in cus:
    m = key == c
    s, e = t[m, 0], t[m, 15]
    span = e.max() - s.min()
    conc.append((e - s).sum() / span); spans.append(span)
    es = np.sort(e); ss = np.sort(s)
    nxt = np.searchsorted(ss, es, side="left")
    ok = nxt < len(ss)
    g = ss[nxt[ok]] - es[ok]
    gaps.extend(g[g < 20000].tolist())
print("per CU: span (cycles/100) median %.1f [min %.1f max %.1f]; time-averaged resident waves median %.2f [min %.2f max %.2f]" % (
    np.median(spans) / 100, min(spans) / 100, max(spans) / 100, np.median(conc), min(conc), max(conc)))
print("end of a wave -> next start on the same CU (cycles/100): median %.2f p90 %.2f" % tuple(np.percentile(gaps, [50, 90]) / 100))
print("waves per SIMD id:", np.bincount(simd))
