#!/bin/bash
# k_poly against the outputs per thread and tile (debug build: HIPSOXR_DEBUG_POLY_R caps R): two-stage job time, three ratios
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HIPSOXR_LIBRARY=$PWD/python-soxr_amd/_variants/dbg/libhipsoxr.so
for r in 0 4 6 8 9 10 12; do
  echo -n "R<=$r: "
  HIPSOXR_DEBUG_POLY_R=$r python - <<'PY'
import sys
sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
out = []
for a, b, fr, ch in ((48000, 44101, 2880000, 2), (44101, 48000, 2880000, 2), (44100, 16001, 2880000, 2), (48000, 44101, 2880000, 1), (48000, 44101, 960000, 2), (48000, 44101, 480000, 1)):
    plan = dev.Plan(a, b, "VHQ")
    x = torch.randn((fr, ch), device="cuda") * 0.25
    if ch == 1: x = x[:, 0].contiguous()
    y = dev.resample_tensor(plan, x)
    job = dev.PreparedJob(plan, x, y)
    for _ in range(5): job.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): job.launch()
    e1.record(); torch.cuda.synchronize()
    out.append("%d->%d/%dk/%dch %.1f" % (a, b, fr // 1000, ch, e0.elapsed_time(e1) * 10))
print("  ".join(out))
PY
done
