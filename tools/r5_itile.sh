#!/bin/bash
# k_interp_tile with two outputs per lane (channel pairs / the column's two halves) against one (HIPSOXR_NO_INTERP_PAIR=1, debug build):
# bit-identity of the results, then launch times of the exact engine on interpolated-phase plans
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_random_rates.py tests/test_gpu_vr.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
export HIPSOXR_LIBRARY=$PWD/python-soxr_amd/_variants/dbg/libhipsoxr.so
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, subprocess, sys, pickle
code = r'''
import sys, pickle
sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
res = {}
g = torch.Generator(device="cuda"); g.manual_seed(3)
for a, b, q in ((48000, 44101, "VHQ"), (44101, 48000, "HQ"), (44100, 16001, "VHQ"), (22050.5, 48000, "VHQ")):
    plan = dev.Plan(a, b, q)
    for ch in (1, 2, 3, 8):
        for frames in (30000, 100003, 700001):
            for dt in (torch.float32, torch.int16, torch.float64):
                x = torch.randn((frames, ch), device="cuda", generator=g) * 0.25
                x = (x * 20000).to(dt) if dt == torch.int16 else x.to(dt)
                for lay in ("inter", "planar"):
                    t = x if lay == "inter" else x.t().contiguous().t()
                    if ch == 1: t = x[:, 0].contiguous()
                    y = dev.resample_tensor(plan, t, kernel=dev.KERNEL_EXACT, dither=(dt == torch.int16))
                    res[(a, b, q, ch, frames, str(dt), lay)] = y.cpu()
pickle.dump(res, open(sys.argv[1], "wb"))
'''
open("/tmp/itile_cmp.py", "w").write(code)
env = dict(os.environ)
env["HIPSOXR_DEBUG_INTERP_PAIR_ALWAYS"] = "1"   # (also where the cost model would keep one output per lane)
subprocess.check_call([sys.executable, "/tmp/itile_cmp.py", "/tmp/itile_pair.pkl"], env=env)
env["HIPSOXR_DEBUG_INTERP_NO_TWIN"] = "1"       # (float pairs on one copy of the span)
subprocess.check_call([sys.executable, "/tmp/itile_cmp.py", "/tmp/itile_pair1.pkl"], env=env)
del env["HIPSOXR_DEBUG_INTERP_PAIR_ALWAYS"], env["HIPSOXR_DEBUG_INTERP_NO_TWIN"]
env["HIPSOXR_NO_INTERP_PAIR"] = "1"
subprocess.check_call([sys.executable, "/tmp/itile_cmp.py", "/tmp/itile_single.pkl"], env=env)
import torch
A, A1, B = pickle.load(open("/tmp/itile_pair.pkl", "rb")), pickle.load(open("/tmp/itile_pair1.pkl", "rb")), pickle.load(open("/tmp/itile_single.pkl", "rb"))
bad = [k for k in A if not (torch.equal(A[k], B[k]) and torch.equal(A1[k], B[k]))]
print("pair (two copies / one copy of the span) vs single: %d cases, %d differ" % (len(A), len(bad)), bad[:5])
PY
for rep in 1 2; do
for v in pair pair1 single; do
  unset HIPSOXR_NO_INTERP_PAIR HIPSOXR_DEBUG_INTERP_NO_TWIN
  if [ $v = single ]; then export HIPSOXR_NO_INTERP_PAIR=1; fi
  if [ $v = pair1 ]; then export HIPSOXR_DEBUG_INTERP_NO_TWIN=1; fi
  echo -n "[$v] "
  python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, "python-soxr_amd")
import torch
from soxr_amd import device as dev
out = []
for a, b, fr, ch, dt in ((48000, 44101, 2880000, 2, torch.float32), (48000, 44101, 2880000, 1, torch.float32), (48000, 44101, 2880000, 2, torch.int16), (48000, 44101, 2880000, 8, torch.float32),
                         (48000, 44101, 2880000, 3, torch.float32), (44100, 16001, 2880000, 2, torch.float32), (48000, 44101, 480000, 2, torch.float32), (48000, 44101, 2880000, 2, torch.float64)):
    plan = dev.Plan(a, b, "VHQ")
    x = torch.randn((fr, ch), device="cuda") * 0.25
    x = (x * 20000).to(dt) if dt == torch.int16 else x.to(dt)
    if ch == 1: x = x[:, 0].contiguous()
    y = dev.resample_tensor(plan, x, kernel=dev.KERNEL_EXACT)
    job = dev.PreparedJob(plan, x, y, kernel=dev.KERNEL_EXACT)
    for _ in range(3): job.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): job.launch()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    out.append("%d->%d/%dk/%dch/%s %.0f" % (a, b, fr // 1000, ch, str(dt).split(".")[1], best))
print("  ".join(out))
PY
done
done
} 2>&1 | tee gpurun_out/r5_itile.txt
