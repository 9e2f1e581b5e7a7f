#!/bin/bash
# k_poly / k_poly2 run length R: "fill the last round" rule against the previous rule (largest R that fits; _variants/oldr), job us by length
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in new oldr; do
  if [ $lib = new ]; then unset HIPSOXR_LIBRARY; else export HIPSOXR_LIBRARY=$PWD/python-soxr_amd/_variants/$lib/libhipsoxr.so; fi
  python - $lib <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
for a, b, ch in ((48000, 44101, 2), (48000, 44101, 1), (44101, 48000, 2), (44100, 16001, 1)):
    plan = dev.Plan(a, b, "VHQ")
    out = []
    for sec in (5, 10, 20, 30, 45, 60, 90, 120, 240):
        x = torch.randn((int(a * sec), ch), device="cuda") * 0.25
        if ch == 1: x = x[:, 0].contiguous()
        y = dev.resample_tensor(plan, x)
        job = dev.PreparedJob(plan, x, y)
        for _ in range(5): job.launch()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): job.launch()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 20)
        out.append("%ds %.1f" % (sec, best))
    print("[%s] %d->%d x%d: " % (sys.argv[1], a, b, ch) + "  ".join(out))
PY
done
done
} 2>&1 | tee gpurun_out/r5_poly_rule.txt
