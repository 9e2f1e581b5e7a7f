#!/usr/bin/env python
"""Debug aid: one configs[2] launch (44.1k -> 16k VHQ, 60 s x 8 channels interleaved) of the strided / frame kernel built
with -DFFT2_TRACE, HIPSOXR_DEBUG_TRACE set; per-wave s_memtime stamps summarised like tools/trace_pair2.py.
Stamp index: 0 start | forward: 1 pass-1 done (input arrived + butterfly + LDS stores), 2 barrier, 3 pass-2, 4 barrier,
5 pass-3 | 6 barrier | inverse: 7, 8, 9, 10, 11 (11 = outputs issued) | 13 HW_ID | 14 XCC_ID | 15 end.
    HIPSOXR_VARIANT=trace HIPSOXR_EXTRA_FLAGS="-DFFT2_TRACE -DHIPSOXR_DEBUG_SWITCHES" bash python-soxr_amd/build.sh   (here)
    HIPSOXR_LIBRARY=python-soxr_amd/_variants/trace/libhipsoxr.so python tools/trace_c2.py [waves per workgroup]          (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
os.environ["HIPSOXR_DEBUG_TRACE"] = "/tmp/hipsoxr_trace.bin"
import torch
from soxr_amd import device as dev
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 5
plan = dev.Plan(44100, 16000, "VHQ")
x = torch.randn((2646000, 8), device="cuda") * 0.25
for _ in range(3):
    y = dev.resample_tensor(plan, x)
    torch.cuda.synchronize()
t = np.fromfile("/tmp/hipsoxr_trace.bin", dtype=np.uint64).reshape(-1, NW, 16).astype(np.int64)
t = t[t[:, 0, 0] != 0]                     # (padded workgroups leave at once and stamp nothing)
start, end = t[:, :, 0], t[:, :, 15]
print("workgroups", t.shape[0])
life = end.max(axis=1) - start.min(axis=1)
print("workgroup lifetime (cycles) median %.0f  p10 %.0f  p90 %.0f" % tuple(np.percentile(life, [50, 10, 90])))
names = ["F1(load+bfly+st)", "bar", "F2", "bar", "F3", "bar", "I1", "bar", "I2", "bar", "I3+store", "tail"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 15]
for w in range(NW):
    seg = np.diff(t[:, w, idx], axis=1)
    print("wave %d median cycles:" % w, " ".join("%s=%.0f" % (n, np.median(seg[:, i])) for i, n in enumerate(names)))
hw, xcc = t[:, 0, 13], t[:, 0, 14] & 0xF
key = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)
ws, we = start.min(axis=1), end.max(axis=1)
cus = np.unique(key)
conc, per_cu, gaps = [], [], []
for c in cus:
    m = key == c
    s, e = np.sort(ws[m]), np.sort(we[m])
    span = e.max() - s.min()
    conc.append((we[m] - ws[m]).sum() / span); per_cu.append(m.sum())
    nxt = np.searchsorted(s, e, side="left"); ok = nxt < len(s)
    g = s[nxt[ok]] - e[ok]; gaps.extend(g[g < 20000].tolist())
print("CUs seen %d; workgroups per CU min %d median %d max %d" % (len(cus), min(per_cu), int(np.median(per_cu)), max(per_cu)))
print("per CU: time-averaged resident workgroups median %.2f [min %.2f max %.2f]" % (np.median(conc), min(conc), max(conc)))
if gaps: print("end of a workgroup -> next start on the same CU (cycles): median %.0f p90 %.0f" % tuple(np.percentile(gaps, [50, 90])))
span_all = [we[key == c].max() - ws[key == c].min() for c in cus]
print("per CU span (cycles) median %.0f max %.0f" % (np.median(span_all), max(span_all)))
