"""Back-to-back launch floor of a trivial kernel on this stack (tools/ubench/libstream_probe.so, 4 KB and 1 MB read sweeps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libstream_probe.so"))
probe.stream_probe_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
a = torch.zeros(1 << 20, device='cuda'); b = torch.zeros(1 << 20, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for nbytes in (4096, 1 << 20):
    for _ in range(5): probe.stream_probe_run(8, b.data_ptr(), a.data_ptr(), nbytes, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): probe.stream_probe_run(8, b.data_ptr(), a.data_ptr(), nbytes, st)
    e1.record(); torch.cuda.synchronize()
    print(nbytes, "bytes read kernel: %.2f us per launch" % (e0.elapsed_time(e1) * 1e3 / 200))
