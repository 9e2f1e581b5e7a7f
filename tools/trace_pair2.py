#!/usr/bin/env python
"""Debug aid: one batch launch of k_fft_pair2 built with -DFFT2_TRACE and HIPSOXR_DEBUG_TRACE set; summarises
the per-wave s_memtime stamps.  Stamp index: 0 start | forward: 1 pass-1 done (input arrived + butterfly +
LDS stores), 2 barrier, 3 pass-2 done, 4 barrier, 5 pass-3 done | 6 barrier | inverse: 7, 8, 9, 10, 11 likewise
(11 = staging written) | 12 barrier | 13 HW_ID | 14 XCC_ID | 15 end (run stored).
    HIPSOXR_VARIANT=trace HIPSOXR_EXTRA_FLAGS="-DFFT2_TRACE -DHIPSOXR_DEBUG_SWITCHES" bash python-soxr_amd/build.sh   (here)
    tools/with_variant.sh trace python tools/trace_pair2.py                                 (GPU box)
Besides the per-phase medians it rebuilds every CU's timeline from the HW_ID / XCC_ID words: how many workgroups
a CU holds over time, how long a freed slot stays empty, and what a CU's steady-state rate is."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
os.environ["HIPSOXR_DEBUG_TRACE"] = "/tmp/hipsoxr_trace.bin"
import torch
from soxr_amd import device as dev
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 6
plan = dev.Plan(48000, 44100, "VHQ")
x = torch.randn((clips, frames, 1), device="cuda") * 0.25
for _ in range(3):
    y = dev.resample_tensor(plan, x)
    torch.cuda.synchronize()
t = np.fromfile("/tmp/hipsoxr_trace.bin", dtype=np.uint64).reshape(-1, NW, 16).astype(np.int64)
start, end = t[:, :, 0], t[:, :, 15]
# s_memtime counts shader cycles; the columns below are cycles / 100 (so 21.0 = 2100 cycles = 1 us at 2.1 GHz).
# Counters of different XCDs are not synchronised: spans across XCDs mean nothing, lifetimes and per-CU timelines do.
print("workgroups", t.shape[0])
life = end.max(axis=1) - start.min(axis=1)
print("workgroup lifetime (cycles/100) median %.2f  p10 %.2f  p90 %.2f" % tuple(np.percentile(life, [50, 10, 90]) / 100))
names = ["F1(load+bfly+st)", "bar", "F2", "bar", "F3", "bar", "I1", "bar", "I2", "bar", "I3+stage", "bar", "store-out"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15]
if os.environ.get("DIF"):  # -DFFT_DIF -DFFT2_TRACE build: stamps 0 start, 1 first pass + barrier, 2 local passes + barrier, 3 inverse first pass + barrier, 4 local + staging + barrier, 15 end
    names = ["F1", "bar", "F-local", "bar", "I1(incl. its barrier)", "bar", "I-local(+barrier+staging)", "bar", "store-out"]
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 15]
mid = t[t.shape[0] // 4: 3 * t.shape[0] // 4]          # steady state: the middle half of the launch
for w in range(NW):
    seg = np.diff(mid[:, w, idx], axis=1)
    print("wave %d median cycles/100:" % w, " ".join("%s=%.2f" % (n, np.median(seg[:, i]) / 100) for i, n in enumerate(names)))

# ---- per-CU timelines -------------------------------------------------------------------------------------
hw, xcc = t[:, 0, 13], t[:, 0, 14] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ws, we = start.min(axis=1), end.max(axis=1)
cus = np.unique(key)
print("distinct (xcc, se, sh, cu) seen: %d; workgroups per CU: min %d median %d max %d" % (
    len(cus), *np.percentile(np.bincount(np.searchsorted(cus, key)), [0, 50, 100]).astype(int)))
conc_all, gap_all, span_all, rate_all, first_all = [], [], [], [], []
for c in cus:
    m = key == c
    s, e = ws[m], we[m]
    o = np.argsort(s)
    s, e = s[o], e[o]
    span = e.max() - s.min()
    busy = (e - s).sum()
    conc_all.append(busy / span)                      # time-averaged number of resident workgroups
    span_all.append(span)
    rate_all.append(span / len(s))                    # cycles per workgroup on this CU
    # how long does a slot stay empty?  pair each end with the next start after it (greedy, in time order)
    es = np.sort(e)
    nxt = np.searchsorted(s, es, side="left")
    ok = nxt < len(s)
    gaps = s[nxt[ok]] - es[ok]
    # only the ends that are followed by a start within the launch (drop the final drain)
    gap_all.extend(gaps[gaps < 20000].tolist())
    first_all.append(np.sort(s)[:4].max() - s.min())  # spread of the first four starts (initial fill)
print("per CU: span (cycles/100) median %.1f  [min %.1f max %.1f]" % (np.median(span_all) / 100, min(span_all) / 100, max(span_all) / 100))
print("per CU: time-averaged resident workgroups median %.2f  [min %.2f max %.2f]" % (np.median(conc_all), min(conc_all), max(conc_all)))
print("per CU: cycles/100 per workgroup (span / count) median %.2f" % (np.median(rate_all) / 100))
print("end of a workgroup -> next start on the same CU (cycles/100): median %.2f  p10 %.2f  p90 %.2f" % tuple(np.percentile(gap_all, [50, 10, 90]) / 100))
print("initial fill: spread of a CU's first four starts (cycles/100) median %.2f" % (np.median(first_all) / 100))
# the drain: how long is a CU below 4 / below 2 resident workgroups at the end of its span
tail4, tail2 = [], []
for c in cus:
    m = key == c
    e = np.sort(we[m])
    if len(e) >= 4:
        tail4.append(e[-1] - e[-4]); tail2.append(e[-1] - e[-2])
print("drain: last end minus 4th-last end per CU (cycles/100) median %.1f; minus 2nd-last %.1f" % (np.median(tail4) / 100, np.median(tail2) / 100))

# ---- do the workgroups of a CU run in step?  At any instant, how many of a CU's resident workgroups are in their load
# phase (stamp 0 -> 1 of wave 0: input requested ... first pass stored)?  If phases were independent the count would be
# binomial with p = the load phase's share of a lifetime; bunching shows as excess mass at 0 and at 3-4.
f1s, f1e = t[:, 0, 0], t[:, 0, 1]
hist = np.zeros(8); res_hist = np.zeros(8); tot = 0.0
for c in cus:
    m = key == c
    ev = []
    for a, b in zip(f1s[m], f1e[m]): ev.append((a, 1, 0)); ev.append((b, -1, 0))
    for a, b in zip(ws[m], we[m]): ev.append((a, 0, 1)); ev.append((b, 0, -1))
    ev.sort()
    lo, hi = np.percentile(ws[m], 20), np.percentile(we[m], 80)     # steady state of this CU
    nl = nr = 0; last = None
    for tt, dl, dr in ev:
        if last is not None and tt > last:
            a, b = max(last, lo), min(tt, hi)
            if b > a:
                hist[min(nl, 7)] += b - a; res_hist[min(nr, 7)] += b - a; tot += b - a
        nl += dl; nr += dr; last = tt
p_load = float(np.median((f1e - f1s) / np.maximum(we - ws, 1)))
print("share of a lifetime in the load phase (median): %.3f" % p_load)
print("time share with k workgroups of the CU in their load phase: " + " ".join("k=%d: %.3f" % (k, hist[k] / tot) for k in range(6)))
print("time share with k workgroups resident:                      " + " ".join("k=%d: %.3f" % (k, res_hist[k] / tot) for k in range(6)))
from math import comb
nres = sum(k * res_hist[k] for k in range(8)) / tot
n = int(round(nres))
print("binomial(n=%d, p=%.3f) for comparison:                        " % (n, p_load) + " ".join("k=%d: %.3f" % (k, comb(n, k) * p_load ** k * (1 - p_load) ** (n - k)) for k in range(min(n, 5) + 1)))
