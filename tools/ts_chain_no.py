#!/usr/bin/env python
"""Per-call time of device-chunk stream calls (soxr_amd.device.TensorStream, int16 44.1k -> 16k VHQ mono) on small chunks;
run under HIPSOXR_DEBUG_CHAIN_NO=<n> with the debug-switch build to vary k_chain's outputs per workgroup."""
import sys, time, os
sys.path.insert(0, "python-soxr_amd")
import numpy as np, torch
from soxr_amd import device as dev
x = (torch.randn(44100 * 20, device="cuda") * 5000).to(torch.int16)
for chunk in (441, 1500, 4410):
    ts = dev.TensorStream(44100, 16000, 1, dtype=torch.int16, quality="VHQ")
    ts.resample_chunk(x[:chunk]); ts.clear(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        ts.clear(); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for a in range(0, len(x) - chunk, chunk):
            ts.resample_chunk(x[a:a + chunk]); n += 1
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    print(f"chunk {chunk}: {best:.1f} us per call", flush=True)
