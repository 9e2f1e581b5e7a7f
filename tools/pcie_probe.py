#!/usr/bin/env python
"""What the host link and the host copies deliver on this box, in the shapes dist._HostPipe uses: pinned <-> device copies of
64 MB blocks (one direction, both at once on two streams), and N threads copying pageable clips into a pinned slot."""
import time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
B, NB = 16 << 20, 30                    # float32 elements per block (64 MB), blocks
pin = [torch.empty(B, dtype=torch.float32, pin_memory=True) for _ in range(3)]
pout = [torch.empty(B, dtype=torch.float32, pin_memory=True) for _ in range(3)]
dev = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(3)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


def h2d():
    with torch.cuda.stream(s1):
        for k in range(NB):
            dev[k % 3].copy_(pin[k % 3], non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        for k in range(NB):
            pout[k % 3].copy_(dev[k % 3], non_blocking=True)


def both():
    h2d(); d2h()


gb = B * 4 * NB / 1e9
for name, fn in (("H2D", h2d), ("D2H", d2h), ("both", both)):
    t = timed(fn)
    print("%-5s %.2f GB each way in %.1f ms = %.1f GB/s per direction" % (name, gb, t * 1e3, gb / t))
src = [np.random.default_rng(i).standard_normal(480000).astype(np.float32) for i in range(34)]
for th in (1, 2, 4, 8, 12):
    pool = ThreadPoolExecutor(th)
    view = pin[0].numpy()

    def put(r):
        for i in r:
            np.copyto(view[i * 480000:(i + 1) * 480000], src[i])

    def stage():
        for _ in range(NB):
            for f in [pool.submit(put, range(t, 34, th)) for t in range(th)]:
                f.result()
    t0 = time.perf_counter(); stage(); t = time.perf_counter() - t0
    print("%2d threads: %d blocks of 34 clips staged in %.1f ms = %.1f GB/s" % (th, NB, t * 1e3, NB * 34 * 1.92e-3 / t))
