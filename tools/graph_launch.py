import sys; sys.path.insert(0,'python-soxr_amd')
import torch
from soxr_amd import device as dev
plan = dev.Plan(48000, 44100, "VHQ")
x = torch.randn(2880000, device="cuda") * 0.25
y = dev.resample_tensor(plan, x)
job = dev.PreparedJob(plan, x, y)
for _ in range(10): job.launch()
torch.cuda.synchronize()
K=200
def timeit(fn):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/K
print("plain launches: %.2f us/step"%timeit(lambda: [job.launch() for _ in range(K)]))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    job_s = dev.PreparedJob(plan, x, y)
    for _ in range(3): job_s.launch()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g, stream=s):
    job_g = dev.PreparedJob(plan, x, y)
    for _ in range(K): job_g.launch()
g.replay(); torch.cuda.synchronize()
print("graph of %d launches: %.2f us/step"%(K, timeit(lambda: g.replay())))
y2 = y.clone(); y.zero_(); g.replay(); torch.cuda.synchronize(); print("same result:", torch.equal(y, y2))
