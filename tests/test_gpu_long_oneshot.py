"""GPU: long clips (tens of seconds, megabytes each way) through soxr.resample / hipsoxr_oneshot — one copy in, one
launch, one copy out — equal the chunked stream driver (the reference's csoxr_divide_proc shape,
src/soxr_ext.cpp:210-273) bit for bit, and the oracle on windows at the head, inside and at the tail; also from several
threads at once (the reference releases the GIL around its drivers, src/soxr_ext.cpp:222,297).
(Round 4 built a pipelined form — the clip in 1.25-4 MiB pieces, a helper thread copying finished outputs back on a
second HIP stream while the next piece goes in — and measured it SLOWER below 120 s: 60 s mono 0.59 vs 0.48 ms; copies
from and to pageable memory pay their page pinning per call, and one big copy each way already runs at the link's
one-way rate.  Removed; profiles/NOTES_r04.md §6.  These tests stay: they are the long-clip parity cases.)"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sig(rng, shape, dtype):
    if np.issubdtype(dtype, np.integer):
        return (rng.standard_normal(shape) * 5000).astype(dtype)
    return (rng.standard_normal(shape) * 0.25).astype(dtype)


@pytest.mark.parametrize("dtype,shape,rates,q", [
    (np.float32, (48000 * 60,), (48000, 44100), "VHQ"),        # configs[1]: 11.5 MB in, 8 pieces
    (np.int16, (44100 * 40, 2), (44100, 16000), "VHQ"),        # 7 MB in, 2.6 MB out
    (np.float64, (480000, 2), (48000, 96000), "HQ"),           # up-sampling: more out than in
    (np.int32, (1300000,), (44100, 48000), "HQ"),
    (np.float32, (900000,), (48000, 44101.5), "HQ"),           # interpolated-phase plan
])
def test_long_oneshot_equals_the_stream_driver(soxr, dtype, shape, rates, q):
    rng = np.random.default_rng(5)
    x = _sig(rng, shape, dtype)
    y = soxr.resample(x, rates[0], rates[1], quality=q)
    want = soxr._resample_divided(x, rates[0], rates[1], quality=q, div_frames=250001)  # one handle fed in pieces, flushed
    assert y.dtype == x.dtype and y.shape == want.shape
    assert np.array_equal(y, want)


def test_long_oneshot_windows_vs_oracle(soxr, oracle):
    rng = np.random.default_rng(6)
    x = _sig(rng, (48000 * 30,), np.float32)
    y = soxr.resample(x, 48000, 44100, quality="VHQ")
    n_out = len(y)
    assert n_out == int(len(x) * 44100 / 48000 + 0.5)
    # head, tail, and windows inside
    piece = max(len(x) * 4 // 8, 5 << 18) // 4
    for a in [0, len(x) - 40000] + [c * piece - 20000 for c in range(1, (len(x) + piece - 1) // piece)]:
        a = max(0, min(a, len(x) - 40000))
        # outputs whose windows lie inside x[a - 400 : a + 40000 + 400) computed by the oracle on that slice
        lo, hi = max(0, a - 1000), min(len(x), a + 41000)
        k0 = -(-lo * 147 // 160) + 400
        k1 = hi * 147 // 160 - 400
        ref = oracle.resample(x[lo:hi], 48000, 44100, "VHQ", mode="port")
        off = lo * 147 / 160
        if lo % 160 == 0:  # the slice starts on a period boundary: same phases
            k_off = lo // 160 * 147
            assert np.array_equal(y[k0:k1], ref[k0 - k_off:k1 - k_off]), a


def test_long_oneshot_from_several_threads(soxr):
    """Results do not depend on what other threads are doing."""
    rng = np.random.default_rng(7)
    xs = [_sig(rng, (48000 * 25,), np.float32) for _ in range(4)]
    want = [soxr._resample_divided(x, 48000, 44100, quality="VHQ", div_frames=300000) for x in xs]
    got = [None] * 4

    def work(i):
        for _ in range(3):
            got[i] = soxr.resample(xs[i], 48000, 44100, quality="VHQ")
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(4):
        assert np.array_equal(got[i], want[i]), i
