"""Frequency-domain engine (HIPSOXR_KERNEL_FFT): same filter as the direct form, evaluated by
overlap-save FFTs.  It is not bit-identical to the canonical order, so the bar here is the
north-star tolerance: <= 1e-6 relative RMS against the oracle's float64 reference (measured:
~1.5e-7 for k_fft_block, ~2.2e-7 for the paired-block kernel), plus a max-error bound, exact lengths, and agreement with the exact engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FFT, EXACT = 5, 6


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


@pytest.mark.parametrize("in_rate,out_rate", [(48000, 44100), (44100, 16000), (44100, 32000), (32000, 44100),
                                              (48000, 22050), (8000, 48000), (44100, 22050), (22050, 32000),
                                              (44100, 48000), (16000, 44100), (96000, 48000), (48000, 96000)])
@pytest.mark.parametrize("quality", ["VHQ", "HQ", "MQ", "LQ"])
def test_fft_engine_within_tolerance_of_oracle(oracle, in_rate, out_rate, quality):
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((30000, 2)) * 0.25).astype(np.float32)
    plan = dev.Plan(in_rate, out_rate, quality)
    xt = torch.from_numpy(x).cuda()
    try:
        y = dev.resample_tensor(plan, xt, kernel=FFT).cpu().numpy()
    except RuntimeError as e:
        if quality in ("MQ", "LQ"):     # 104 dB stop band: its aliasing is above the 1e-6 bar
            assert "FFT engine needs" in str(e)
            return
        assert "FFT engine unavailable" in str(e)
        pytest.skip("no 7-smooth block for this plan")
    assert quality in ("VHQ", "HQ")
    ref = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
    assert y.shape == ref.shape
    err = y.astype(np.float64) - ref
    assert _rms(err) / _rms(ref) <= 1e-6
    assert np.abs(err).max() <= 1e-5 * _rms(ref) * 4
    exact = dev.resample_tensor(plan, xt, kernel=EXACT).cpu().numpy()
    assert _rms(y.astype(np.float64) - exact) / _rms(exact) <= 1e-6


@pytest.mark.parametrize("length", [1, 7, 100, 4703, 4704, 4705, 9000, 100001])
def test_fft_engine_lengths_and_edges(oracle, length):
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(length)
    x = (rng.standard_normal(length) * 0.25).astype(np.float32)
    plan = dev.Plan(48000, 44100, "VHQ")
    y = dev.resample_tensor(plan, torch.from_numpy(x).cuda(), kernel=FFT).cpu().numpy()
    ref = oracle.resample(x, 48000, 44100, "VHQ", mode="ref")
    assert y.shape == ref.shape
    assert _rms(y - ref) <= 1e-6 * max(_rms(ref), 1e-3)


def test_fft_engine_batch_and_strides(oracle):
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((3, 20000, 4)) * 0.25).astype(np.float32)
    plan = dev.Plan(48000, 44100, "VHQ")
    xt = torch.from_numpy(x).cuda()
    y = dev.resample_tensor(plan, xt, kernel=FFT).cpu().numpy()
    for clip in range(3):
        ref = oracle.resample(x[clip], 48000, 44100, "VHQ", mode="ref")
        assert _rms(y[clip] - ref) / _rms(ref) <= 1e-6
    # planar (channel-major) view of the same data
    xp = xt.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    yp = dev.resample_tensor(plan, xp, kernel=FFT).cpu().numpy()
    # the interleaved layout pairs channels, the planar one pairs blocks: same filter, different
    # partner in the complex transform -> equal to rounding, not bit for bit
    assert _rms(yp - y) <= 5e-7 * _rms(y)
    assert np.array_equal(dev.resample_tensor(plan, xp, kernel=FFT).cpu().numpy(), yp)   # deterministic


@pytest.mark.parametrize("in_rate,out_rate,frames", [(44100, 48000, 300000), (44100, 16000, 400000),
                                                     (16000, 44100, 150000), (96000, 48000, 500000),
                                                     (48000, 96000, 250000), (48000, 44100, 300000)])
def test_paired_kernel_families_at_size(oracle, in_rate, out_rate, frames):
    """Every ratio with a compile-time paired-block schedule, at a size where AUTO uses it (several
    hundred blocks, odd block count, edge blocks at both ends), two channels interleaved."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(frames)
    x = (rng.standard_normal((frames + 13, 2)) * 0.25).astype(np.float32)
    plan = dev.Plan(in_rate, out_rate, "VHQ")
    y = dev.resample_tensor(plan, torch.from_numpy(x).cuda()).cpu().numpy()        # AUTO
    ref = oracle.resample(x, in_rate, out_rate, "VHQ", mode="ref")
    assert y.shape == ref.shape
    err = y.astype(np.float64) - ref
    assert _rms(err) / _rms(ref) <= 1e-6
    # no block seam stands out: the error of every 4096-sample stretch stays at the same level
    seg = np.sqrt(np.mean(err[: len(err) // 4096 * 4096].reshape(-1, 4096, 2) ** 2, axis=(1, 2)))
    assert seg.max() <= 4e-6 * _rms(ref)


@pytest.mark.parametrize("in_rate,out_rate", [(48000, 16000), (16000, 48000), (48000, 32000), (32000, 48000),
                                              (192000, 48000), (48000, 192000), (48000, 8000), (8000, 48000),
                                              (44100, 32000), (32000, 44100), (88200, 48000), (48000, 88200),
                                              (96000, 44100), (44100, 96000), (44100, 8000), (8000, 44100),
                                              (192000, 44100), (44100, 192000), (22050, 32000), (32000, 22050), (44100, 12000), (12000, 44100), (24000, 32000), (32000, 24000)])
def test_paired_kernel_schedule_table(oracle, in_rate, out_rate):
    """Every further ratio with a compile-time schedule (hipsoxr::launch_fft `pairs` table), HQ and
    VHQ, mono (block pairing) and stereo interleaved (channel pairing), against the float64 oracle."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(int(in_rate + out_rate))
    frames = int(0.9 * in_rate) + 17
    x = (rng.standard_normal((frames, 2)) * 0.25).astype(np.float32)
    for quality in ("VHQ", "HQ"):
        plan = dev.Plan(in_rate, out_rate, quality)
        ref = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
        y2 = dev.resample_tensor(plan, torch.from_numpy(x).cuda(), kernel=FFT).cpu().numpy()
        y1 = dev.resample_tensor(plan, torch.from_numpy(np.ascontiguousarray(x[:, 0])).cuda(), kernel=FFT).cpu().numpy()
        assert y2.shape == ref.shape
        assert _rms(y2 - ref) <= 1e-6 * _rms(ref)
        assert _rms(y1 - ref[:, 0]) <= 1e-6 * _rms(ref[:, 0])


def test_fft_engine_refuses_what_it_cannot_do():
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(48000, 44100, "VHQ")
    x16 = torch.zeros(1000, dtype=torch.int16, device="cuda")
    with pytest.raises(RuntimeError):
        dev.resample_tensor(plan, x16, kernel=FFT)     # integer I/O stays on the exact engine


def test_host_surface_never_uses_the_fft_engine(soxr, oracle):
    """soxr.resample keeps the bit-exact contract at sizes where AUTO would pick the FFT engine."""
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(600000) * 0.25).astype(np.float32)
    y = soxr.resample(x, 48000, 44100, quality="VHQ")
    pl = oracle.plan(48000, 44100, "VHQ")
    for k0 in (0, 250000, len(y) - 500):
        want = oracle.resample_channel(pl, x, "port_f32", k0=k0, n_out=500)
        assert np.array_equal(y[k0:k0 + 500], want)


@pytest.mark.parametrize("in_rate,out_rate,quality,tol", [(48000, 44100, "VHQ", 2e-9), (44100, 48000, "VHQ", 2e-9),
                                                          (44100, 16000, "VHQ", 2e-9), (48000, 44100, "HQ", 1e-6),
                                                          (96000, 48000, "VHQ", 2e-9), (16000, 48000, "VHQ", 2e-9)])
def test_fft_engine_float64_instance(oracle, in_rate, out_rate, quality, tol):
    """float64 device jobs: the paired kernel in double2 (libsoxr's own VHQ engine is a float64 one, SURVEY.md
    §0.3).  Against the oracle's float64 direct form what is left is the method's own floor — the neglected
    aliasing of the stop band: ~3e-10 relative RMS for VHQ (-177 dB), ~4e-7 for HQ (-128 dB) — not rounding."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(17)
    plan = dev.Plan(in_rate, out_rate, quality)
    for shape in ((50000,), (3, 20001, 1), (7, 1)):            # mono, a batch of planar columns, a tiny job
        x = rng.standard_normal(shape) * 0.25
        xt = torch.from_numpy(x).cuda()
        y = dev.resample_tensor(plan, xt, kernel=FFT)
        assert y.dtype == torch.float64
        y = y.cpu().numpy()
        cols = x.reshape(-1, x.shape[-2] if x.ndim == 3 else x.shape[0]) if x.ndim != 2 else x.T
        if x.ndim == 1:
            ref = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
        elif x.ndim == 3:
            ref = np.stack([oracle.resample(x[c, :, 0], in_rate, out_rate, quality, mode="ref") for c in range(x.shape[0])])[:, :, None]
        else:
            ref = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
        assert y.shape == ref.shape
        assert _rms(y - ref) <= tol * max(_rms(ref), 1e-3), (shape, _rms(y - ref) / max(_rms(ref), 1e-3))


FFT_F64 = 8  # hipsoxr_kernel_t: the frequency-domain engine computing in float64 whatever the I/O type


@pytest.mark.parametrize("in_rate,out_rate,quality", [(48000, 44100, "VHQ"), (44100, 48000, "VHQ"), (44100, 16000, "VHQ"),
                                                      (96000, 48000, "VHQ"), (48000, 44100, "HQ")])
def test_fft_engine_float32_io_on_float64_arithmetic(oracle, in_rate, out_rate, quality):
    """HIPSOXR_KERNEL_FFT_F64 on float32 jobs: loads widen, the whole chain runs in double2, the result is rounded to
    float32 once — the arithmetic width libsoxr's VHQ recipe itself uses for float32 clients (SURVEY.md §0.3;
    reference src/soxr_ext.cpp:74,228).  Against the oracle's float64 direct form (on the oracle's own bank) what is
    left is the float32 OUTPUT rounding, 2^-24/sqrt(3) ~ 3.4e-8 of the sample magnitude: <= 5e-8 relative RMS for VHQ
    (HQ adds its -128 dB stop-band aliasing: <= 1e-6).  The float32-arithmetic kernel sits at ~2e-7 on the same input."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(23)
    plan = dev.Plan(in_rate, out_rate, quality)
    tol = 5e-8 if quality == "VHQ" else 1e-6
    for shape in ((120000,), (3, 30001, 1), (9, 1)):            # mono, a batch of planar columns, a tiny job
        x = (rng.standard_normal(shape) * 0.25).astype(np.float32)
        xt = torch.from_numpy(x).cuda()
        y = dev.resample_tensor(plan, xt, kernel=FFT_F64)
        assert y.dtype == torch.float32
        y = y.cpu().numpy().astype(np.float64)
        if x.ndim == 3:
            ref = np.stack([oracle.resample(x[c, :, 0].astype(np.float64), in_rate, out_rate, quality, mode="ref")
                            for c in range(x.shape[0])])[:, :, None]
        else:
            ref = oracle.resample(x.astype(np.float64), in_rate, out_rate, quality, mode="ref")
        assert y.shape == ref.shape
        assert _rms(y - ref) <= tol * max(_rms(ref), 1e-3), (shape, _rms(y - ref) / max(_rms(ref), 1e-3))
        if x.ndim == 1 and quality == "VHQ":                   # and it really is tighter than the float32-arithmetic kernel
            y32 = dev.resample_tensor(plan, xt, kernel=FFT).cpu().numpy().astype(np.float64)
            assert _rms(y - ref) < 0.5 * _rms(y32 - ref)
    # float64 jobs take the same kernel id (they run float64 arithmetic anyway); interleaved float32 data is refused
    xd = torch.from_numpy(rng.standard_normal(50000) * 0.25).cuda()
    assert torch.equal(dev.resample_tensor(plan, xd, kernel=FFT_F64), dev.resample_tensor(plan, xd, kernel=FFT))
    with pytest.raises(RuntimeError):
        dev.resample_tensor(plan, torch.zeros((40000, 2), dtype=torch.float32, device="cuda"), kernel=FFT_F64)


def test_auto_engine_float64_large_job_is_frequency_domain_and_close(oracle):
    """AUTO on a large float64 device job now takes the frequency-domain engine: not bit-identical to the
    canonical order any more (the host surface still is: it passes KERNEL_EXACT), within 2e-9 of it."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(18)
    x = rng.standard_normal(200000) * 0.25
    plan = dev.Plan(48000, 44100, "VHQ")
    xt = torch.from_numpy(x).cuda()
    auto = dev.resample_tensor(plan, xt).cpu().numpy()
    exact = dev.resample_tensor(plan, xt, kernel=EXACT).cpu().numpy()
    assert np.array_equal(exact, oracle.resample(x, 48000, 44100, "VHQ", mode="port"))
    assert 0 < _rms(auto - exact) <= 2e-9 * _rms(exact)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-6), (np.float64, 2e-9)])
@pytest.mark.parametrize("in_rate,out_rate,ch", [(44100, 16000, 8), (48000, 44100, 4), (44100, 48000, 2), (16000, 44100, 6)])
def test_channel_pair_kernel(oracle, dtype, tol, in_rate, out_rate, ch):
    """Interleaved data with an even channel count runs the channel-pair kernel (k_fft_strided2<..., true>: buffer loads and
    stores of one (Real, Real) word per frame, XCD-aware ids), float32 and float64: every channel within the
    engine's tolerance of the oracle's float64 direct form, first and last blocks included, and a strided view
    (channels 2..5 of a wider tensor) gives the same numbers."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(ch * 1000 + in_rate)
    x = (rng.standard_normal((2, 60013, ch)) * 0.25).astype(dtype)
    plan = dev.Plan(in_rate, out_rate, "VHQ")
    xt = torch.from_numpy(x).cuda()
    y = dev.resample_tensor(plan, xt, kernel=FFT).cpu().numpy()
    for clip in range(2):
        ref = oracle.resample(x[clip].astype(np.float64), in_rate, out_rate, "VHQ", mode="ref")
        assert y[clip].shape == ref.shape
        for c in range(ch):
            assert _rms(y[clip][:, c] - ref[:, c]) / _rms(ref[:, c]) <= tol
    if ch >= 6:
        view = xt[:, :, 2:6]                                   # frame stride ch, 4 channels, 8/16-byte aligned start
        yv = dev.resample_tensor(plan, view, kernel=FFT).cpu().numpy()
        assert np.array_equal(yv, y[:, :, 2:6])
    if ch >= 6:
        view = xt[:, :, 1:5]                                   # pairs (1,2), (3,4): words aligned to the element only
        yv = dev.resample_tensor(plan, view, kernel=FFT).cpu().numpy()
        for clip in range(2):
            ref = oracle.resample(x[clip][:, 1:5].astype(np.float64), in_rate, out_rate, "VHQ", mode="ref")
            assert _rms(yv[clip] - ref) / _rms(ref) <= tol


@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-6), (np.float64, 2e-9)])
@pytest.mark.parametrize("in_rate,out_rate,ch", [(48000, 44100, 3), (44100, 16000, 5), (44100, 48000, 1)])
def test_strided_column_kernel(oracle, dtype, tol, in_rate, out_rate, ch):
    """Columns with a frame stride that cannot be paired by channel — odd channel counts of interleaved data, a
    single channel sliced out of a wider tensor — run the strided second-generation kernel (two blocks of one column
    per transform): every channel within tolerance of the oracle, float32 and float64."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(ch * 77 + in_rate)
    wide = (rng.standard_normal((2, 50021, ch + 2)) * 0.25).astype(dtype)
    plan = dev.Plan(in_rate, out_rate, "VHQ")
    wt = torch.from_numpy(wide).cuda()
    for view, ref_in in ((wt[:, :, :ch], wide[:, :, :ch]), (wt[:, :, 2:2 + ch], wide[:, :, 2:2 + ch])):
        y = dev.resample_tensor(plan, view, kernel=FFT).cpu().numpy()          # frame stride ch + 2
        for clip in range(2):
            ref = oracle.resample(ref_in[clip].astype(np.float64), in_rate, out_rate, "VHQ", mode="ref")
            assert y[clip].shape == ref.shape
            for c in range(ch):
                assert _rms(y[clip][:, c] - ref[:, c]) / _rms(ref[:, c]) <= tol
    xc = torch.from_numpy(np.ascontiguousarray(wide[:, :, :ch])).cuda()          # frame stride ch (odd)
    if ch > 1:
        yc = dev.resample_tensor(plan, xc, kernel=FFT).cpu().numpy()
        assert _rms(yc - dev.resample_tensor(plan, wt[:, :, :ch], kernel=FFT).cpu().numpy()) <= 5e-7 * _rms(yc)


@pytest.mark.parametrize("in_rate,out_rate,frames", [(48000, 44100, 96000), (48000, 44100, 1300000), (44100, 48000, 700000)])
@pytest.mark.parametrize("quality", ["VHQ", "HQ"])
def test_quarter_size_blocks(oracle, in_rate, out_rate, frames, quality):
    """Jobs of a few hundred block pairs run quarter-size blocks (1280/1176 points): same bar against the oracle,
    first and last outputs included, and no block seam stands out."""
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(frames)
    x = (rng.standard_normal(frames + 11) * 0.25).astype(np.float32)
    plan = dev.Plan(in_rate, out_rate, quality)
    y = dev.resample_tensor(plan, torch.from_numpy(x).cuda()).cpu().numpy()        # AUTO
    ref = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
    assert y.shape == ref.shape
    err = y.astype(np.float64) - ref
    assert _rms(err) / _rms(ref) <= 1e-6
    seg = np.sqrt(np.mean(err[: len(err) // 2048 * 2048].reshape(-1, 2048) ** 2, axis=1))
    assert seg.max() <= 4e-6 * _rms(ref)
    assert abs(err[:500]).max() <= 1e-5 and abs(err[-500:]).max() <= 1e-5
