#!/bin/bash
# two-stage form: "FFT 1:2 first" for mild down-sampling (round 5) against round 4's order, on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
D=$R/python-soxr_amd/_variants/dbg/libhipsoxr.so
timeout 900 python -m pytest tests/test_gpu_two_stage.py tests/test_gpu_random_rates.py -x -q -m gpu 2>&1 | tail -3
cat > /tmp/ts_time.py <<'PY'
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
for (a, b, ch) in ((48000, 44101, 2), (48000, 44101, 1), (96000, 88201, 2), (48000, 32001, 2), (44100, 40000.5, 2), (48000, 36003, 2)):
    plan = dev.Plan(a, b, "VHQ")
    x = torch.randn((int(a) * 60, ch), device="cuda") * 0.25
    y = dev.resample_tensor(plan, x)
    job = dev.PreparedJob(plan, x, y)
    for _ in range(5): job.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): job.launch()
    e1.record(); torch.cuda.synchronize()
    ye = dev.resample_tensor(plan, x[:480000], kernel=dev.KERNEL_EXACT).double()
    ya = dev.resample_tensor(plan, x[:480000]).double()
    print("%g -> %g x%d: %.1f us   rel err vs exact %.2e" % (a, b, ch, e0.elapsed_time(e1) * 1e3 / 50, float((ya - ye).norm() / ye.norm())))
PY
for rep in 1 2; do
echo "[up-first <= 1.5]"; HIPSOXR_LIBRARY=$D timeout 300 python /tmp/ts_time.py 2>/dev/null
echo "[round-4 order]";   HIPSOXR_LIBRARY=$D HIPSOXR_DEBUG_TS_UP_LIMIT=100 timeout 300 python /tmp/ts_time.py 2>/dev/null
done
