import sys, time
sys.path.insert(0, "python-soxr_amd")
import numpy as np, soxr_amd as soxr
rng = np.random.default_rng(5)
x = (rng.standard_normal(44100 * 20) * 5000).astype(np.int16)
for vr in (False, True):
    rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ", vr=vr)
    rs.resample_chunk(x[:96000]); rs.clear()
    t0 = time.perf_counter(); n = 0
    for a in range(0, len(x), 96000):
        if vr and n == 2: rs.set_io_ratio(44100, 22050, 1000)
        rs.resample_chunk(x[a:a + 96000], last=(a + 96000 >= len(x))); n += 1
    print("vr" if vr else "cr", (time.perf_counter() - t0) / n * 1e6, "us per call")
