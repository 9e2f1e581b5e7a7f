#!/bin/bash
# configs[2]: parity of the frames kernel, then frames vs channel-pair kernel interleaved on one box, then its trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_size.py tests/test_gpu_tensor_stream.py -x -q -m gpu 2>&1 | tail -4
D=$R/python-soxr_amd/_variants/dbg/libhipsoxr.so
for rep in 1 2 3; do
  echo -n "[frames rot3] "; HIPSOXR_LIBRARY=$D ROTATE=3 timeout 120 python tools/run_workload.py c2 300 2>/dev/null | tail -n 1
  echo -n "[pairs  rot3] "; HIPSOXR_LIBRARY=$D HIPSOXR_FFT_NO_FRAMES=1 ROTATE=3 timeout 120 python tools/run_workload.py c2 300 2>/dev/null | tail -n 1
done
HIPSOXR_LIBRARY=python-soxr_amd/_variants/trace/libhipsoxr.so timeout 120 python tools/trace_c2.py 8 2>&1 | tail -24
timeout 300 python - <<'PY'
import sys, torch, time
sys.path.insert(0, "."); sys.path.insert(0, "python-soxr_amd")
from soxr_amd import device as dev
nch = 128
xm = (torch.randn((441 * 300, nch), device="cuda") * 5000).to(torch.int16)
grp = dev.TensorStreamGroup(nch, 44100, 16000, 1, dtype=torch.int16, quality="VHQ", dither_seeds=list(range(nch)))
for i, s in enumerate(grp.streams):
    s.resample_chunk(xm[: 7 * i, 0].contiguous())
xg = xm.t().contiguous()
chunks = [xg[:, a:a + 441].contiguous() for a in range(0, xg.shape[1], 441)]
grp.resample_chunks(chunks[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
for c in chunks[1:]:
    grp.resample_chunks(c)
torch.cuda.synchronize()
print("128 independent handles x 441 frames: %.2f us per call" % ((time.perf_counter() - t0) / (len(chunks) - 1) * 1e6))
ts = dev.TensorStream(44100, 16000, 1, dtype=torch.int16, quality="VHQ")
x1 = xm[:, 0].contiguous()
ts.resample_chunk(x1[:441]); torch.cuda.synchronize()
parts = [x1[a:a + 441] for a in range(441, len(x1), 441)]
t0 = time.perf_counter()
for c in parts:
    ts.resample_chunk(c)
torch.cuda.synchronize()
print("one handle x 441 frames: %.2f us per call" % ((time.perf_counter() - t0) / len(parts) * 1e6))
PY
