#!/bin/bash
# k_poly2 (channel pairs of interleaved data in one pass) against k_poly (one channel per pass): parity, then two-stage job times
# A/B, interleaved (debug build: HIPSOXR_POLY_NO_PAIR=1 keeps k_poly)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_two_stage.py tests/test_gpu_random_rates.py -q -x 2>&1 | tail -3
python tests/fuzz/fuzz_two_stage.py 150 7 2>&1 | tail -2
export HIPSOXR_LIBRARY=$PWD/python-soxr_amd/_variants/dbg/libhipsoxr.so
for rep in 1 2 3; do
for v in pair single; do
  if [ $v = single ]; then export HIPSOXR_POLY_NO_PAIR=1; else unset HIPSOXR_POLY_NO_PAIR; fi
  echo -n "[$v] "
  python - <<'PY'
import sys
sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
out = []
for a, b, fr, ch, q in ((48000, 44101, 2880000, 2, "VHQ"), (44101, 48000, 2880000, 2, "VHQ"), (44100, 16001, 2880000, 2, "VHQ"), (48000, 44101, 2880000, 8, "VHQ"),
                        (48000, 44101, 2880000, 2, "HQ"), (96000, 88201, 5760000, 2, "VHQ"), (48000, 44101, 480000, 2, "VHQ"),
                        (48000, 44101, 2880000, 1, "VHQ"), (44101, 48000, 2880000, 1, "VHQ"), (44100, 16001, 2880000, 1, "VHQ"), (48000, 44101, 2880000, 3, "VHQ"), (48000, 44101, 2880000, -2, "VHQ")):
    plan = dev.Plan(a, b, q)
    x = torch.randn((fr, abs(ch)), device="cuda") * 0.25
    if ch == 1: x = x[:, 0].contiguous()
    if ch < 0: x = x.t().contiguous().t()        # planar
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
    out.append("%d->%d/%s/%dk/%dch %.1f" % (a, b, q, fr // 1000, ch, best))
print("  ".join(out))
PY
done
done
} 2>&1 | tee gpurun_out/r5_poly2.txt
