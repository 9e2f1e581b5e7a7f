"""GPU: variable-rate streams (ResampleStream(vr=True) + set_io_ratio) — reference API
src/soxr/__init__.py:80-82, :162-179; src/soxr_ext.cpp:74, :200-204; usage tests/vr.py:60-114.

Parity: bit-exact against the oracle's variable-rate evaluation driven by an independent
integer restatement of the clock (tests/vr_sim.py), for every dtype; plus the analytic property
(sine on a moving clock) on the product itself and the API's error behaviour.
"""
import numpy as np
import pytest

from vr_sim import VrSim

pytestmark = pytest.mark.gpu


def _signal(rng, n, dtype):
    if np.issubdtype(dtype, np.integer):
        return (rng.standard_normal(n) * 5000).astype(dtype)
    return (rng.standard_normal(n) * 0.25).astype(dtype)


SCHEDULE = [  # (frames fed, then: set_io_ratio(in, out, slew) or None)
    (4800, None), (0, None), (4801, (5, 2, 1500)), (9000, None), (123, (3, 1, 0)), (7000, (2, 1, 400)),
    (300, (9, 4, 250)),   # changes ratio again while the previous slew is still running
    (6000, None), (5000, None)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
@pytest.mark.parametrize("quality", ["HQ", "VHQ", "QQ"])
def test_vr_stream_bit_exact_vs_oracle(soxr, oracle, dtype, quality):
    rng = np.random.default_rng(42)
    rs = soxr.ResampleStream(48000, 16000, 1, dtype=dtype, quality=quality, vr=True)
    sim = VrSim(oracle, 48000, 16000, quality, dtype)
    total = 0
    for i, (n, change) in enumerate(SCHEDULE):
        x = _signal(rng, n, dtype)
        last = i == len(SCHEDULE) - 1
        y = rs.resample_chunk(x, last=last)
        want = sim.feed(x, last=last)
        assert y.dtype == np.dtype(dtype)
        assert len(y) == len(want), f"chunk {i}"
        assert np.array_equal(y, want), f"chunk {i}"
        total += len(y)
        if change:
            rs.set_io_ratio(*change)
            sim.set_io_ratio(change[0] / change[1], change[2])
    assert total > 10000
    assert rs.delay() < 2


LARGE = [  # chunks of >= 4096 outputs (k_interp_wave): generic steps, a slew across launches, steps of exactly 2 and 1
    (30000, None), (20000, (44100, 22050, 3000)), (30000, None), (24001, (5, 2, 0)), (96000, (1, 1, 500)), (20000, (44100, 16000, 0)),
    (40000, None)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
@pytest.mark.parametrize("quality", ["VHQ", "HQ", "LQ"])
def test_vr_large_chunks_bit_exact_vs_oracle(soxr, oracle, dtype, quality):
    """Round 4: launches of 4096 outputs and more run k_interp_wave (an output's two half-chains on two lanes, cubic
    records fetched coalesced and transposed through LDS).  Same arithmetic per output as before: every chunk equals the
    oracle driven by the independent clock of tests/vr_sim.py bit for bit — for steps that spread the outputs over all
    phase intervals, for a step of exactly 2.0 / 1.0 (every output in ONE interval) and for slews that run across calls."""
    rng = np.random.default_rng(11)
    rs = soxr.ResampleStream(44100, 16000, 1, dtype=dtype, quality=quality, vr=True)
    sim = VrSim(oracle, 44100, 16000, quality, dtype)
    total = 0
    for i, (n, change) in enumerate(LARGE):
        x = _signal(rng, n, dtype)
        last = i == len(LARGE) - 1
        y = rs.resample_chunk(x, last=last)
        want = sim.feed(x, last=last)
        assert y.dtype == np.dtype(dtype) and len(y) == len(want), f"chunk {i}"
        assert np.array_equal(y, want), f"chunk {i}"
        total += len(y)
        if change:
            rs.set_io_ratio(change[0], change[1], change[2])
            sim.set_io_ratio(change[0] / change[1], change[2])
    assert total > 100000


def test_vr_large_chunk_channels_share_the_kernel(soxr, oracle):
    """Interleaved channels of a large variable-rate chunk: each column equals the mono oracle."""
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((50000, 4)) * 5000).astype(np.int16)
    rs = soxr.ResampleStream(44100, 16000, 4, dtype="int16", quality="VHQ", vr=True)
    rs.set_io_ratio(44100, 20000, 0)
    y = rs.resample_chunk(x, last=True)
    for c in range(4):
        sim = VrSim(oracle, 44100, 16000, "VHQ", np.int16)
        sim.set_io_ratio(44100 / 20000, 0)
        assert np.array_equal(y[:, c], sim.feed(x[:, c], last=True, channel=c)), c
    assert y.shape[1] == 4 and abs(len(y) - 50000 * 20000 / 44100) < 3


@pytest.mark.parametrize("chunk", [441, 4410])
def test_configs4_variable_rate_int16_44k1_to_16k_chunked(soxr, oracle, chunk):
    """BASELINE configs[4]: ResampleStream variable-rate 44100 -> 16000 int16, chunked input, state
    carried on the device across calls; the ratio is moved mid-stream and back."""
    rng = np.random.default_rng(chunk)
    x = (rng.standard_normal(44100) * 5000).astype(np.int16)
    rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ", vr=True)
    sim = VrSim(oracle, 44100, 16000, "VHQ", np.int16)
    got, want = [], []
    for i in range(0, len(x), chunk):
        last = i + chunk >= len(x)
        if i == 5 * chunk:
            rs.set_io_ratio(44100, 22050, 300)
            sim.set_io_ratio(44100 / 22050, 300)
        if i == 8 * chunk:
            rs.set_io_ratio(44100, 16000, 0)
            sim.set_io_ratio(44100 / 16000, 0)
        got.append(rs.resample_chunk(x[i:i + chunk], last=last))
        want.append(sim.feed(x[i:i + chunk], last=last))
        assert len(got[-1]) == len(want[-1])
    got, want = np.concatenate(got), np.concatenate(want)
    assert got.dtype == np.int16 and np.array_equal(got, want)
    assert rs.num_clips() == 0


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.float64])
@pytest.mark.parametrize("chunk", [100, 441, 1500])
def test_vr_stream_through_the_resident_kernel(soxr, oracle, dtype, chunk):
    """Variable-rate streams on the resident kernel (round 3: every message carries its own Q64.64 clock — position, step,
    step increment — in nine more tagged words): the same frames in the same calls as the oracle driven by the independent
    clock of tests/vr_sim.py, across ratio changes with and without a slew, a change while a slew is running, and idle
    gaps longer than the kernel's idle time (the instance leaves and the next call starts another)."""
    import time
    rng = np.random.default_rng(7 + chunk)
    x = _signal(rng, 30000, dtype)
    rs = soxr.ResampleStream(44100, 16000, 1, dtype=dtype, quality="VHQ", vr=True, resident=True)
    sim = VrSim(oracle, 44100, 16000, "VHQ", dtype)
    changes = {3: (44100, 22050, 300), 5: (5, 2, 0), 9: (44100, 30000, 1000), 10: (44100, 16000, 50)}
    for c, i in enumerate(range(0, len(x), chunk)):
        last = i + chunk >= len(x)
        if c in changes:
            a, b, slew = changes[c]
            rs.set_io_ratio(a, b, slew)
            sim.set_io_ratio(a / b, slew)
        if c == 7:
            time.sleep(0.005)                                   # the idle instance leaves (1 ms)
        y = rs.resample_chunk(x[i:i + chunk], last=last)
        want = sim.feed(x[i:i + chunk], last=last)
        assert y.dtype == np.dtype(dtype) and len(y) == len(want), f"call {c}"
        assert np.array_equal(y, want), f"call {c}"
    assert rs.delay() < 2


def test_vr_multichannel_and_chunking_of_calls(soxr, oracle):
    """Channels share the clock; cutting the same input into different process calls between the
    same ratio changes gives the same samples."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((30000, 3)) * 0.25).astype(np.float32)

    def run(cuts):
        rs = soxr.ResampleStream(44100, 22050, 3, dtype="float32", quality="HQ", vr=True)
        out, pos = [], 0
        for seg, change in [(12000, (3, 2, 1000)), (18000, None)]:
            end = pos + seg
            for a in range(pos, end, cuts):
                b = min(a + cuts, end)
                out.append(rs.resample_chunk(x[a:b], last=(b == len(x))))
            pos = end
            if change:
                # apply the change at a known output index: drain first so that nothing is pending
                rs.set_io_ratio(*change)
        return np.concatenate(out)

    a = run(30000)
    b = run(997)
    n = min(len(a), len(b))
    # the change takes effect at the output index reached when it is requested; with different
    # cuts that index differs by less than one chunk's look-ahead, so compare the common prefix
    # before the change and the overall length
    first = 3000
    assert np.array_equal(a[:first], b[:first])
    assert abs(len(a) - len(b)) <= 2
    assert a.shape[1] == 3
    sim = VrSim(oracle, 44100, 22050, "HQ", np.float32)
    w0 = sim.feed(x[:12000, 1])
    assert np.array_equal(a[:len(w0), 1], w0)


def test_vr_sine_follows_the_clock(soxr):
    fs = 48000.0
    rs = soxr.ResampleStream(fs, 24000.0, 1, dtype="float64", quality="VHQ", vr=True)
    n = np.arange(48000)
    x = np.sin(2 * np.pi * 1000.0 / fs * n)
    y0 = rs.resample_chunk(x[:24000])
    rs.set_io_ratio(1.0, 1.0, 0)                  # from 2:1 to 1:1 at once
    y1 = rs.resample_chunk(x[24000:], last=True)
    # first part: plain 2:1 decimation of the sine
    k = np.arange(len(y0))
    assert np.abs(y0[200:] - np.sin(2 * np.pi * 1000.0 / fs * 2 * k)[200:]).max() < 1e-6
    # second part: every input sample once, continuing from position 2*len(y0)
    t = 2 * len(y0) + np.arange(len(y1))
    inside = t < len(x) - 400
    assert np.abs(y1[inside] - np.sin(2 * np.pi * 1000.0 / fs * t[inside])).max() < 1e-6
    assert abs(len(y0) * 2 + len(y1) - len(x)) <= 2


def test_vr_api_errors(soxr):
    rs = soxr.ResampleStream(48000, 16000, 1, vr=True)
    with pytest.raises(RuntimeError):
        rs.set_io_ratio(4, 1)                     # beyond the largest ratio given at construction
    with pytest.raises(ValueError):
        rs.set_io_ratio(0, 1)
    rs.set_io_ratio(3, 1)
    rs.set_io_ratio(1, 1, 100)
    fixed = soxr.ResampleStream(48000, 16000, 1)
    with pytest.raises(RuntimeError):
        fixed.set_io_ratio(2, 1)                  # needs vr=True
    assert "hip" in rs.engine()
    # clear() keeps the last requested ratio and starts a fresh signal
    x = np.zeros(4000, np.float32)
    a = rs.resample_chunk(x, last=True)
    rs.clear()
    b = rs.resample_chunk(x, last=True)
    assert abs(len(b) - 4000) <= 1 and len(a) > 1300
