"""Design-independent parity (SURVEY.md §7.3-1, §8d "band-limited variant … for design-independent
parity").

libsoxr is absent, so the filter design cannot be compared with libsoxr's.  What CAN be shown:

1. The product's host design (plan.cpp, through the C ABI) equals the oracle's design
   (oracle/design.py: numpy/scipy, different ratio reduction, Bessel function, sinc, layout and
   normalisation code) to ~1e-13 of the coefficient scale — two implementations, not one twice.
2. On BAND-LIMITED input any two filters that meet the recipe's pass/stop/attenuation spec must
   produce the same output up to the pass-band ripple (VHQ 2^-27, HQ 2^-19).  So white noise
   low-passed to < 0.85 x Nyquist is resampled (a) by the oracle and (b) on the GPU by both engines,
   and compared with scipy.signal.resample_poly driven by a filter that scipy DESIGNED
   (kaiserord + firwin from the recipe's numbers; none of our design code involved) at
   <= 1e-6 relative RMS.  That is the design-independent half of the north star's 1e-6 claim.
"""
import numpy as np
import pytest
from scipy import signal

RATES = [(48000, 44100), (44100, 16000), (44100, 32000), (32000, 44100), (48000, 22050), (8000, 48000),
         (44100, 22050), (22050, 32000), (100, 200), (48000, 24000), (96000, 44100), (100.5, 200)]
QUALS = ["VHQ", "HQ", "MQ", "LQ", "QQ"]


@pytest.mark.parametrize("in_rate,out_rate", RATES)
@pytest.mark.parametrize("quality", QUALS)
def test_product_bank_vs_independent_design(oracle, in_rate, out_rate, quality):
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality)
    o = oracle.plan(in_rate, out_rate, quality)          # oracle/design.py
    assert (p.L, p.M, p.taps, p.phases) == (o.L, o.M, o.T, o.phases)
    pb = p.bank()
    assert pb.shape == o.bank.shape
    assert np.abs(pb - o.bank).max() <= 2e-13 * np.abs(o.bank).max()


@pytest.mark.parametrize("in_rate,out_rate,quality", [(48000, 44101, "VHQ"), (44100.123456789, 47999.987654321, "HQ"),
                                                       (48000, 44101.5, "HQ"), (12345, 54321.5, "MQ"),
                                                       (44100.37, 48001.11, "QQ"), (48000.77 * 1.0234567, 44100, "MQ")])
def test_product_interp_table_vs_independent_design(oracle, in_rate, out_rate, quality):
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality)
    o = oracle.plan(in_rate, out_rate, quality)
    assert (p.L, p.M, p.taps, p.phases) == (o.L, o.M, o.T, o.phases) and p.phases > 0
    assert np.abs(p.bank() - o.bank).max() <= 1e-12 * np.abs(o.bank).max()


@pytest.mark.parametrize("in_rate,out_rate,quality", [(48000, 16000, "HQ"), (44100, 16000, "VHQ"), (48000, 24000, "QQ")])
def test_product_vr_table_vs_independent_design(oracle, in_rate, out_rate, quality):
    from soxr_amd import device as dev
    p = dev.Plan(in_rate, out_rate, quality, vr=True)
    v = oracle.VrPlan(in_rate, out_rate, quality)
    assert (p.taps, p.phases) == (v.T, v.phases)
    assert np.abs(p.bank() - v.bank).max() <= 1e-12 * np.abs(v.bank).max()


def test_ratio_reduction_random_floats(oracle):
    """Exact-rational continued fraction (design.py) == the product's double-precision one."""
    from soxr_amd import device as dev
    from oracle import design
    rng = np.random.default_rng(7)
    for _ in range(3000):
        a, b = (float(rng.uniform(4000, 200000)) for _ in range(2))
        if rng.integers(3) == 0:
            a = float(round(a))
        if rng.integers(3) == 0:
            b = round(b * 2) / 2
        p = dev.Plan(a, b, "QQ")
        assert (p.L, p.M) == design.ratio(a, b), (a, b)


# ---- band-limited experiment -------------------------------------------------------------------
def scipy_designed_filter(in_rate, out_rate, L, M, bits, pb):
    """A spec-compliant prototype at rate L*in_rate designed ENTIRELY by scipy: kaiserord from the
    recipe's attenuation and transition band, firwin for the windowed sinc."""
    fn = 0.5 * min(in_rate, out_rate)
    nyq_hi = 0.5 * L * in_rate
    att = (bits + 1) * 20 * np.log10(2) + 3.0
    width = (1.0 - pb) * fn / nyq_hi
    numtaps, beta = signal.kaiserord(att, width)
    numtaps |= 1                                     # odd length: integer group delay, zero-latency alignment
    h = signal.firwin(numtaps, 0.5 * (1.0 + pb) * fn / nyq_hi, window=("kaiser", beta))
    return h                                        # (resample_poly applies the gain L itself)


def band_limited_noise(n, frac, seed):
    """White Gaussian noise low-passed (brick wall, in the FFT domain) to frac x Nyquist, with a
    raised-cosine fade at both ends so that the zero-padded edges carry no wide-band energy."""
    rng = np.random.default_rng(seed)
    X = np.fft.rfft(rng.standard_normal(n))
    X[int(frac * (n // 2)):] = 0
    x = np.fft.irfft(X, n)
    fade = min(4096, n // 8)
    w = 0.5 - 0.5 * np.cos(np.pi * np.arange(fade) / fade)
    x[:fade] *= w
    x[-fade:] *= w[::-1]
    return 0.25 * x / x.std()


CASES = [(48000, 44100, "VHQ", 28), (48000, 44100, "HQ", 20), (44100, 16000, "VHQ", 28), (44100, 48000, "VHQ", 28)]


def reference_by_scipy(x64, in_rate, out_rate, quality, bits, oracle):
    o = oracle.plan(in_rate, out_rate, quality)
    h = scipy_designed_filter(in_rate, out_rate, o.L, o.M, bits, oracle.quality(quality)[1])
    y = signal.resample_poly(x64, o.L, o.M, window=h)          # zero-phase, zero-extended: same alignment
    return y[:o.out_len(len(x64))]


def rel_rms(a, b):
    n = min(len(a), len(b))
    return float(np.sqrt(np.mean((a[:n].astype(np.float64) - b[:n]) ** 2)) / np.sqrt(np.mean(b[:n] ** 2)))


@pytest.mark.parametrize("in_rate,out_rate,quality,bits", CASES)
def test_oracle_vs_scipy_designed_filter_band_limited(oracle, in_rate, out_rate, quality, bits):
    x = band_limited_noise(4 * in_rate, 0.85 * min(1.0, out_rate / in_rate), seed=11)
    ref = reference_by_scipy(x, in_rate, out_rate, quality, bits, oracle)
    y = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
    assert abs(len(y) - len(ref)) <= 1
    err = rel_rms(y, ref)
    assert err <= (1e-7 if bits >= 28 else 1e-5), err   # pass-band ripple of two different designs


@pytest.mark.gpu
@pytest.mark.parametrize("in_rate,out_rate,quality,bits", CASES)
@pytest.mark.parametrize("kernel", ["exact", "fft"])
def test_gpu_vs_scipy_designed_filter_band_limited(oracle, in_rate, out_rate, quality, bits, kernel):
    """C1-sized for the headline ratio (60 s), both engines, float32 I/O: <= 1e-6 relative RMS against
    a filter none of our code designed."""
    import torch
    from soxr_amd import device as dev
    seconds = 60 if (in_rate, out_rate, quality) == (48000, 44100, "VHQ") else 8
    x = band_limited_noise(seconds * in_rate, 0.85 * min(1.0, out_rate / in_rate), seed=12)
    ref = reference_by_scipy(x, in_rate, out_rate, quality, bits, oracle)
    plan = dev.Plan(in_rate, out_rate, quality)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    y = dev.resample_tensor(plan, xt, kernel=dev.KERNEL_EXACT if kernel == "exact" else dev.KERNEL_FFT)
    torch.cuda.synchronize()
    err = rel_rms(y.cpu().numpy(), ref)
    assert err <= (1e-6 if bits >= 28 else 1e-5), err


# ---- round 3: the same experiment wider — float64 I/O at a 1e-8 bar, the 16-bit recipes against their own ripple,
#      configs[2]'s shape, up-sampling at size, float32 I/O on float64 arithmetic ------------------------------------
# (in_rate, out_rate, recipe, bits, share of the lower Nyquist the input occupies, bar for float64 results)
WIDE = [(48000, 44100, "VHQ", 28, 0.85, 1e-8), (44100, 48000, "VHQ", 28, 0.85, 1e-8), (16000, 44100, "VHQ", 28, 0.85, 1e-8),
        (96000, 44100, "VHQ", 28, 0.85, 1e-8), (44100, 16000, "HQ", 20, 0.85, 1e-6),
        (48000, 44100, "MQ", 16, 0.85, 5e-6), (44100, 48000, "MQ", 16, 0.85, 5e-6),
        (48000, 44100, "LQ", 16, 0.60, 2e-5)]   # LQ's pass band ends at 0.676 x Nyquist: the input stays inside it


@pytest.mark.parametrize("in_rate,out_rate,quality,bits,frac,bar", WIDE)
def test_oracle_float64_vs_scipy_designed_filter_wide(oracle, in_rate, out_rate, quality, bits, frac, bar):
    """The oracle's float64 direct form against a filter scipy designed from the recipe's numbers alone: two designs that
    meet the same spec agree inside the pass band to the recipe's ripple (VHQ ~3e-10, HQ ~6e-8, MQ ~9e-7, LQ ~5e-6
    measured)."""
    x = band_limited_noise(4 * in_rate, frac * min(1.0, out_rate / in_rate), seed=13)
    ref = reference_by_scipy(x, in_rate, out_rate, quality, bits, oracle)
    y = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
    assert abs(len(y) - len(ref)) <= 1
    assert rel_rms(y, ref) <= bar, rel_rms(y, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("in_rate,out_rate,quality,bits,frac,bar", WIDE)
@pytest.mark.parametrize("kernel", ["exact", "fft"])
def test_gpu_float64_vs_scipy_designed_filter(oracle, in_rate, out_rate, quality, bits, frac, bar, kernel):
    """float64 I/O on the GPU — the canonical-order float64 engine, and the frequency-domain engine's float64 instance
    where the recipe admits it (HQ/VHQ) — against the scipy-designed filter at the float64 bars above (1e-8 for VHQ):
    the float32 cases of round 2 could not see below 5e-8."""
    import torch
    from soxr_amd import device as dev
    if kernel == "fft" and bits < 20:
        pytest.skip("MQ/LQ are not admitted to the frequency-domain engine (104 dB stop band)")
    seconds = 20 if (in_rate, out_rate) in ((44100, 48000), (16000, 44100)) else 6       # up-sampling at size
    x = band_limited_noise(seconds * in_rate, frac * min(1.0, out_rate / in_rate), seed=14)
    ref = reference_by_scipy(x, in_rate, out_rate, quality, bits, oracle)
    plan = dev.Plan(in_rate, out_rate, quality)
    y = dev.resample_tensor(plan, torch.from_numpy(x).cuda(), kernel=dev.KERNEL_EXACT if kernel == "exact" else dev.KERNEL_FFT)
    assert y.dtype == torch.float64
    assert rel_rms(y.cpu().numpy(), ref) <= bar, rel_rms(y.cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("in_rate,out_rate,quality,bits", [(48000, 44100, "VHQ", 28), (44100, 48000, "VHQ", 28)])
def test_gpu_float32_io_float64_arithmetic_vs_scipy_designed_filter(oracle, in_rate, out_rate, quality, bits):
    """HIPSOXR_KERNEL_FFT_F64 (float32 I/O, float64 arithmetic: libsoxr's own VHQ width) against the scipy-designed
    filter: what is left is the float32 rounding of the result, <= 5e-8 — the float32-arithmetic kernels sit at
    5e-8 (canonical order) to 2e-7 (frequency domain) on the same input."""
    import torch
    from soxr_amd import device as dev
    x = band_limited_noise(20 * in_rate, 0.85 * min(1.0, out_rate / in_rate), seed=15)
    ref = reference_by_scipy(x, in_rate, out_rate, quality, bits, oracle)
    plan = dev.Plan(in_rate, out_rate, quality)
    y = dev.resample_tensor(plan, torch.from_numpy(x.astype(np.float32)).cuda(), kernel=dev.KERNEL_FFT_F64)
    assert y.dtype == torch.float32
    # (the reference is computed from the float64 signal; the float32 rounding of the INPUT is part of the error: ~3e-8 more)
    assert rel_rms(y.cpu().numpy(), ref) <= 7e-8, rel_rms(y.cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["exact", "auto"])
def test_gpu_configs2_shape_vs_scipy_designed_filter(oracle, kernel):
    """configs[2]'s shape — [frames, 8] interleaved float32, VHQ 44.1k -> 16k, 20 s — every channel against the
    scipy-designed filter (AUTO = the channel-pair frequency-domain kernel)."""
    import torch
    from soxr_amd import device as dev
    in_rate, out_rate, ch = 44100, 16000, 8
    x = np.stack([band_limited_noise(20 * in_rate, 0.85 * out_rate / in_rate, seed=20 + c) for c in range(ch)], axis=1)
    plan = dev.Plan(in_rate, out_rate, "VHQ")
    y = dev.resample_tensor(plan, torch.from_numpy(x.astype(np.float32)).cuda(),
                            kernel=dev.KERNEL_EXACT if kernel == "exact" else dev.KERNEL_AUTO).cpu().numpy()
    assert y.shape == (plan.out_len(x.shape[0]), ch)
    for c in range(ch):
        ref = reference_by_scipy(x[:, c], in_rate, out_rate, "VHQ", 28, oracle)
        assert rel_rms(y[:, c], ref) <= 1e-6, (c, rel_rms(y[:, c], ref))
