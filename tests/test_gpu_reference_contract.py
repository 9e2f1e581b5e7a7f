"""The behavioural contract python-soxr's own test-suite pins, re-expressed against
soxr_amd on the GPU: dtype preservation, exact lengths, bit-exact equality of the one-shot, divided,
split-channel and streamed drivers, known-answer tones for all recipes, integer paths within the
reference's 2 LSB, thread safety.  Our own bars are tighter where noted (integer paths are
deterministic here, so they are compared exactly)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def tone2(freq, rate, seconds):
    n = int(rate * seconds)
    s = np.sin(2 * np.pi * freq / rate * np.arange(n)) * np.hanning(n)
    return np.stack([s, np.zeros_like(s)], axis=-1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
def test_dtype_is_preserved(soxr, dtype):
    x = np.random.default_rng(0).standard_normal(100).astype(dtype)
    y = soxr.resample(x, 100, 200)
    assert y.dtype == x.dtype and y.shape == (200,)


def test_list_input_becomes_float32(soxr):
    y = soxr.resample([0.0, 1.0, 0.0, -1.0] * 25, 100, 200)
    assert y.dtype == np.float32 and y.shape == (200,)


@pytest.mark.parametrize("in_rate,out_rate", [(44100, 32000), (32000, 44100)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_drivers_agree_bit_for_bit(soxr, in_rate, out_rate, dtype):
    x = np.random.default_rng(1).standard_normal((25999, 2)).astype(dtype)
    one = soxr._resample_oneshot(x, in_rate, out_rate)
    assert np.array_equal(one, soxr.resample(x, in_rate, out_rate))
    assert np.array_equal(one, soxr.resample(np.asfortranarray(x), in_rate, out_rate))
    assert np.array_equal(one, soxr._resample_divided(x, in_rate, out_rate))               # chunked driver
    assert np.array_equal(one, soxr._resample_divided(np.asfortranarray(x), in_rate, out_rate))
    assert np.array_equal(one, soxr._resample_divided(x, in_rate, out_rate, div_frames=1000))


@pytest.mark.parametrize("in_rate,out_rate", [(44100, 32000), (32000, 44100)])
@pytest.mark.parametrize("length", [0, 1, 2, 99, 100, 101, 31999, 32000, 32001, 34829, 44100, 48001, 66150, 166151])
def test_lengths_and_sliced_inputs(soxr, in_rate, out_rate, length):
    x = np.random.default_rng(2).standard_normal((166151, 2)).astype(np.float32)
    xf = np.asfortranarray(x)
    one = soxr._resample_oneshot(x[:length], in_rate, out_rate)
    assert one.shape == (int(np.floor(length * out_rate / in_rate + 0.5)), 2)
    assert np.array_equal(one, soxr.resample(x[:length], in_rate, out_rate))
    assert np.array_equal(one, soxr.resample(xf[:length], in_rate, out_rate))
    assert np.array_equal(one, soxr._resample_divided(x[:length], in_rate, out_rate))


@pytest.mark.parametrize("channels", [1, 2, 3, 5, 7, 24, 49])
def test_channel_slices(soxr, channels):
    x = np.random.default_rng(3).standard_normal((15013, 49)).astype(np.float32)
    one = soxr._resample_oneshot(x[:, :channels], 44100, 32000)
    assert one.shape[1] == channels
    assert np.array_equal(one, soxr.resample(x[:, :channels], 44100, 32000))
    assert np.array_equal(one, soxr.resample(np.asfortranarray(x)[:, :channels], 44100, 32000))


def _stream(soxr, x, in_rate, out_rate, chunk, dtype, quality="HQ"):
    rs = soxr.ResampleStream(in_rate, out_rate, x.shape[1], dtype=dtype, quality=quality)
    parts = [np.empty((0, x.shape[1]), dtype)]
    if len(x) == 0:
        parts.append(rs.resample_chunk(x, last=True))
    for i in range(0, len(x), chunk):
        parts.append(rs.resample_chunk(x[i:i + chunk], last=(i + chunk >= len(x))))
    return np.concatenate(parts)


@pytest.mark.parametrize("in_rate,out_rate", [(44100, 32000), (32000, 44100)])
@pytest.mark.parametrize("chunk", [17, 509, 44100])
@pytest.mark.parametrize("length", [0, 100, 31999, 44100])
@pytest.mark.parametrize("dtype", ["float32", np.float64])
def test_stream_is_chunk_invariant(soxr, in_rate, out_rate, chunk, length, dtype):
    x = np.random.default_rng(4).standard_normal((length, 1)).astype(dtype)
    assert np.array_equal(soxr._resample_oneshot(x, in_rate, out_rate),
                          _stream(soxr, x, in_rate, out_rate, chunk, np.dtype(dtype)))


@pytest.mark.parametrize("in_rate,out_rate", [(48000, 22050), (8000, 48000)])
@pytest.mark.parametrize("chunk", [50, 101])
@pytest.mark.parametrize("length", [1, 101, 32000, 44101])
@pytest.mark.parametrize("dtype", ["int32", np.int16])
def test_stream_int(soxr, in_rate, out_rate, chunk, length, dtype):
    x = (np.random.default_rng(5).standard_normal((length, 2)) * 5000).astype(dtype)
    one = soxr._resample_oneshot(x, in_rate, out_rate)
    st = _stream(soxr, x, in_rate, out_rate, chunk, np.dtype(dtype))
    assert np.allclose(one, st, atol=2)       # the reference's bar
    assert np.array_equal(one, st)            # ours: dither is position-keyed, so exactly equal


@pytest.mark.parametrize("in_rate,out_rate", [(44100, 22050), (22050, 32000)])
@pytest.mark.parametrize("quality", ["VHQ", "HQ", "SOXR_MQ", "lq", "soxr_qq"])
def test_known_answer_tone(soxr, in_rate, out_rate, quality):
    x, want = tone2(32.0, in_rate, 2.0), tone2(32.0, out_rate, 2.0)
    q = soxr.VHQ if quality == "VHQ" else quality
    for xx in (x, x.astype(np.float32)):
        assert np.allclose(want, soxr.resample(xx, in_rate, out_rate, quality=q), atol=1e-4)
        assert np.allclose(want, soxr.resample(np.asfortranarray(xx), in_rate, out_rate, quality=q), atol=1e-4)


@pytest.mark.parametrize("in_rate,out_rate", [(48000, 24000), (32000, 44100)])
@pytest.mark.parametrize("dtype", [np.int32, np.int16])
def test_known_answer_tone_int(soxr, in_rate, out_rate, dtype):
    x = (tone2(32.0, in_rate, 2.0) * 16384).astype(dtype)
    want = (tone2(32.0, out_rate, 2.0) * 16384).astype(dtype)
    got = soxr.resample(x, in_rate, out_rate)
    assert np.allclose(want, got, atol=2)
    assert np.array_equal(got, soxr.resample(np.asfortranarray(x), in_rate, out_rate))
    assert np.array_equal(got, soxr._resample_oneshot(x, in_rate, out_rate))


@pytest.mark.parametrize("n_tasks", [2, 5, 12, 32])
def test_threads(soxr, n_tasks):
    """Distinct handles are usable concurrently (the reference releases the GIL; ctypes does too)."""
    x = np.random.default_rng(6).standard_normal((75999, 2)).astype(np.float32)
    xi = (np.random.default_rng(7).standard_normal((70001, 2)) * 5000).astype(np.int16)
    with ThreadPoolExecutor() as pool:
        fl = list(pool.map(lambda a: soxr.resample(a, 44100, 32000), [x] * n_tasks))
        it = list(pool.map(lambda a: soxr.resample(a, 32000, 48000), [xi] * n_tasks))
    assert all(np.array_equal(fl[0], r) for r in fl[1:])
    assert all(np.array_equal(it[0], r) for r in it[1:])     # deterministic dither: exactly equal


def test_stream_protocol(soxr):
    rs = soxr.ResampleStream(44100, 16000, 2, dtype="int16", quality="VHQ")
    assert rs.engine().startswith("hip-gfx950")
    assert rs.delay() == 0.0 and rs.num_clips() == 0
    with pytest.raises(TypeError):
        rs.resample_chunk(np.zeros((10, 2), np.float32))        # dtype mismatch
    with pytest.raises(TypeError):
        rs.resample_chunk([[0, 0]])                             # not an ndarray
    with pytest.raises(ValueError):
        rs.resample_chunk(np.zeros((10, 3), np.int16))          # channel mismatch
    with pytest.raises(ValueError):
        rs.resample_chunk(np.zeros((2, 2, 2), np.int16))
    x = (np.random.default_rng(8).standard_normal((10000, 2)) * 3000).astype(np.int16)
    y0 = rs.resample_chunk(x[:5000])
    assert rs.delay() > 0
    y1 = rs.resample_chunk(x[5000:], last=True)
    assert rs.delay() == pytest.approx(0.0, abs=1.0)
    with pytest.raises(RuntimeError):
        rs.resample_chunk(x[:10])                               # input after last input
    full = np.concatenate([y0, y1])
    rs.clear()                                                  # same config, fresh signal
    again = rs.resample_chunk(x, last=True)
    assert np.array_equal(full, again)
    assert np.array_equal(full, soxr.resample(x, 44100, 16000, quality="VHQ"))
    with pytest.raises(RuntimeError):
        rs.set_io_ratio(44100, 22050)                           # needs vr=True (tests/test_gpu_vr.py)


def test_csoxr_handle_view(soxr):
    """`ResampleStream._csoxr` (src/soxr/__init__.py:99): the reference's own conftest calls
    `rs._csoxr.engine()` (tests/conftest.py:7,10); the binding's names and units (src/soxr_ext.cpp:408-423)."""
    import gc
    import weakref
    rs = soxr.ResampleStream(44100, 48000, 1, quality="VHQ")
    c = rs._csoxr
    assert c.engine() == rs.engine() and c.engine().startswith("hip-gfx950")
    assert (c.in_rate, c.out_rate, c.channels, c.ended) == (44100.0, 48000.0, 1, False)
    x = (np.random.default_rng(3).standard_normal(4000) * 0.25).astype(np.float32)
    y = c.process_float32(x[:, None], True)                     # 2-D in, 2-D out, as the binding's process<T>
    assert c.ended and y.ndim == 2 and np.array_equal(y[:, 0], soxr.resample(x, 44100, 48000, quality="VHQ"))
    with pytest.raises(TypeError):
        c.process_int16(x[:, None].astype(np.int16), False)
    c.clear()
    assert not c.ended and c.delay() == 0.0 and c.num_clips() == 0
    with pytest.raises(RuntimeError):
        c.set_io_ratio(2.0)                                     # not a variable-rate stream
    v = soxr.ResampleStream(48000, 24000, 1, quality="HQ", vr=True)
    v._csoxr.set_io_ratio(1.5, 10)                              # io_ratio = in/out, below the constructor's 2.0
    gc.disable()
    try:                                                        # the view holds no strong reference to the stream:
        rs2 = soxr.ResampleStream(44100, 48000, 1)              # device state is released by reference counting alone
        r2 = weakref.ref(rs2)
        del rs2
        assert r2() is None
    finally:
        gc.enable()


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16, np.int32])
@pytest.mark.parametrize("chunk", [17, 441, 4410, 50000])
def test_deferred_stream_equals_oneshot(soxr, dtype, chunk):
    """Deferred output (extension): every call returns the previous call's frames; the concatenation is
    bit-identical to the one-shot result and to the synchronous stream, delay() accounts for what is
    pending, and clear() starts over."""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((44100, 2))
    x = (x * 5000).astype(dtype) if np.issubdtype(dtype, np.integer) else (x * 0.25).astype(dtype)
    want = soxr.resample(x, 44100, 16000, quality="HQ")
    rs = soxr.ResampleStream(44100, 16000, 2, dtype=dtype, quality="HQ", deferred=True)
    for rep in range(2):
        parts, fed, got = [], 0, 0
        for a in range(0, len(x), chunk):
            y = rs.resample_chunk(x[a:a + chunk], last=(a + chunk >= len(x)))
            fed += len(x[a:a + chunk]); got += len(y)
            parts.append(y)
            if a + chunk < len(x):
                assert abs(rs.delay() - (fed * 16000 / 44100 - got)) < 1e-6
        out = np.concatenate(parts)
        assert out.shape == want.shape and np.array_equal(out, want)
        if chunk <= 4410:
            assert len(parts[0]) == 0      # nothing can come back from the first call
        rs.clear()


def test_deferred_stream_small_output_buffer(soxr):
    """The C entry with an output buffer smaller than what is pending: frames are handed out over several
    calls, nothing is lost or reordered."""
    import ctypes as C
    from soxr_amd import _native as nat
    rng = np.random.default_rng(32)
    x = (rng.standard_normal(20000) * 0.25).astype(np.float32)
    want = soxr.resample(x, 48000, 44100, quality="HQ")
    h = C.c_void_p()
    nat.check(nat.lib.hipsoxr_stream_create(48000.0, 44100.0, 1, nat.FLOAT32_I, nat.HQ, nat.DEFER, C.byref(h)))
    out, done = np.zeros(len(want) + 100, np.float32), C.c_size_t()
    pos = 0
    for a in range(0, len(x), 1000):
        c = np.ascontiguousarray(x[a:a + 1000])
        nat.check(nat.lib.hipsoxr_stream_process(h, c.ctypes.data, len(c), out.ctypes.data + 4 * pos, 300, C.byref(done)))
        pos += done.value
    while True:
        nat.check(nat.lib.hipsoxr_stream_process(h, None, 0, out.ctypes.data + 4 * pos, 300, C.byref(done)))
        if done.value == 0:
            break
        pos += done.value
    nat.lib.hipsoxr_stream_delete(h)
    assert pos == len(want) and np.array_equal(out[:pos], want)


@pytest.mark.parametrize("rates,quality", [((44100, 16000), "VHQ"), ((16000, 44100), "VHQ"), ((48000, 44100), "HQ"),
                                           ((44100, 44101), "HQ"), ((8000, 192000), "MQ"), ((192000, 8000), "LQ"),
                                           ((48000, 47999.5), "VHQ")])
def test_stream_backlog_bound(soxr, rates, quality):
    """resample_chunk sizes its output buffer from a bound on the frames a synchronous stream can still owe —
    (taps/2 + 2) * out/in + 1 — instead of asking delay() every call: the bound must hold after every call, for
    exact and interpolated plans, whatever the chunk size (a violated bound would only delay frames, silently)."""
    from soxr_amd import device as dev
    plan = dev.Plan(*rates, quality)
    ratio = rates[1] / rates[0]
    bound = (plan.taps / 2 + 2) * ratio + 1
    rng = np.random.default_rng(7)
    rs = soxr.ResampleStream(*rates, 1, dtype=np.float32, quality=quality)
    fed = 0
    for n in [1, 7, 100, 3, 2000, 441, 1, 1, 5000, 64, 9999]:
        rs.resample_chunk((rng.standard_normal(n) * 0.25).astype(np.float32))
        fed += n
        assert rs.delay() <= bound, (n, fed, rs.delay(), bound)
