#!/bin/bash
# k_poly2 variants: coefficient records in flight (CH), workgroups per CU (launch bound), run length R
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in dbg p2ch8 p2occ4; do
for r in 0 6 4; do
  export HIPSOXR_LIBRARY=$PWD/python-soxr_amd/_variants/$lib/libhipsoxr.so HIPSOXR_DEBUG_POLY_R=$r
  echo -n "[$lib R<=$r] "
  python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
out = []
for a, b, fr, ch, q in ((48000, 44101, 2880000, 2, "VHQ"), (44101, 48000, 2880000, 2, "VHQ"), (44100, 16001, 2880000, 2, "VHQ"), (48000, 44101, 2880000, 2, "HQ")):
    plan = dev.Plan(a, b, q)
    x = torch.randn((fr, ch), device="cuda") * 0.25
    y = dev.resample_tensor(plan, x)
    job = dev.PreparedJob(plan, x, y)
    for _ in range(5): job.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): job.launch()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    out.append("%d->%d/%s %.1f" % (a, b, q, best))
print("  ".join(out))
PY
done
done
done
} 2>&1 | tee gpurun_out/r5_poly2_var.txt
