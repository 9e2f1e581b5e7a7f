import sys; sys.path.insert(0,'python-soxr_amd')
import torch
from soxr_amd import _native as nat
a=torch.zeros(1<<20, device='cuda'); b=torch.zeros(1<<20, device='cuda')
st=torch.cuda.current_stream().cuda_stream
for nbytes in (4096, 1<<20):
    for _ in range(5): nat.lib.hipsoxr_bench_stream(b.data_ptr(), a.data_ptr(), nbytes, 1, st)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): nat.lib.hipsoxr_bench_stream(b.data_ptr(), a.data_ptr(), nbytes, 1, st)
    e1.record(); torch.cuda.synchronize()
    print(nbytes, "bytes read kernel: %.2f us per launch"%(e0.elapsed_time(e1)*1e3/200))
