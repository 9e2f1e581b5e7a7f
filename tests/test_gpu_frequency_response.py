"""What the GPU engines OUTPUT, measured against the recipe's numbers — not against the oracle and not against
`plan.cpp`'s coefficients (SURVEY.md §4 "frequency-response / alias-rejection tests"; the reference's own analytic-tone
contract, tests/test_resample.py:133-156, is the 1e-4 version of check (i)).

For every engine (canonical-order f32 / f64, frequency-domain f32 / f64, float32 I/O on float64 arithmetic), recipe and
rate pair one launch resamples a batch of probe signals, and numpy measures on the results:

  (i)   pass-band tones (12 frequencies up to the recipe's pass-band end): amplitude within the recipe's ripple
        2^-(bits-1) (float32 engines: or their rounding floor, whichever is larger);
  (ii)  what is left after removing the tone from its own output (images, aliases, noise) and, when down-sampling,
        the output of tones between the output Nyquist and the input Nyquist: at most -(bits+1)*6.02 dB of the input
        for the float64 engines (3 dB of measurement margin), at most the float32 rounding floor for float32 ones —
        the frequency-domain engine's "neglected aliasing <= -176 dB" claim is asserted here, on tones;
  (iii) an impulse on a period boundary: the response is symmetric about its own output instant (linear phase).

Only `Plan`'s recipe numbers (precision bits, pass-band end, stop-band begin) and numpy enter the expectations."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATES = [(48000, 44100), (44100, 16000), (44100, 48000)]
# engine: (numpy dtype, kernel id, recipes it serves)
EXACT, FFT, FFT_F64 = 6, 5, 8
ENGINES = {
    "exact_f32": (np.float32, EXACT, ("VHQ", "HQ", "MQ", "LQ")),
    "exact_f64": (np.float64, EXACT, ("VHQ", "HQ", "MQ", "LQ")),
    "fft_f32": (np.float32, FFT, ("VHQ", "HQ")),            # (the 16-bit recipes' 104 dB stop band would miss the 1e-6 class: not admitted)
    "fft_f64": (np.float64, FFT, ("VHQ", "HQ")),
    "fft_f32_on_f64": (np.float32, FFT_F64, ("VHQ", "HQ")),
}
N_PASS, N_STOP = 12, 8
F32_FLOOR_DB = -118.0   # float32 engines: rounding noise of ~300-700 products, relative to a full-scale tone (measured ~ -135 .. -140)
F32_AMP_TOL = 3e-7


def _probes(in_rate, out_rate, plan, frames, dtype):
    """[clips, frames] float64: pass-band tones, stop-band tones (down-sampling only), one impulse; and their frequencies."""
    ny = min(in_rate, out_rate) / 2.0
    t = np.arange(frames) / in_rate
    fp = plan.passband_end * ny * np.array([0.013, 0.11, 0.23, 0.31, 0.42, 0.53, 0.61, 0.72, 0.83, 0.91, 0.96, 0.995])
    sig = [np.cos(2 * np.pi * f * t + 0.3 * i) for i, f in enumerate(fp)]
    fs = np.array([])
    if out_rate < in_rate:
        lo, hi = plan.stopband_begin * ny * 1.004, in_rate / 2.0 * 0.998
        fs = lo + (hi - lo) * np.array([0.0, 0.07, 0.19, 0.33, 0.5, 0.68, 0.86, 1.0])
        sig += [np.cos(2 * np.pi * f * t + 0.2 * i) for i, f in enumerate(fs)]
    imp = np.zeros(frames)
    n0 = (frames // 2) // plan.M * plan.M   # a period boundary: output instant n0 * L / M is an integer
    imp[n0] = 1.0
    sig.append(imp)
    return np.stack(sig).astype(dtype), fp, fs, n0


def _fit(y, f, rate):
    """Least-squares amplitude of the tone at f in the middle of y, and the RMS of what is left."""
    n = len(y)
    a, b = int(0.2 * n), int(0.8 * n)
    k = np.arange(a, b)
    ph = 2 * np.pi * f * k / rate
    basis = np.stack([np.cos(ph), np.sin(ph)], axis=1)
    seg = y[a:b].astype(np.float64)
    coef, *_ = np.linalg.lstsq(basis, seg, rcond=None)
    res = seg - basis @ coef
    return float(np.hypot(*coef)), float(np.sqrt(np.mean(res ** 2)))


@pytest.mark.parametrize("in_rate,out_rate", RATES)
@pytest.mark.parametrize("quality", ["VHQ", "HQ", "MQ", "LQ"])
@pytest.mark.parametrize("engine", list(ENGINES))
def test_engine_output_meets_the_recipe(engine, quality, in_rate, out_rate):
    import torch
    from soxr_amd import device as dev
    dtype, kernel, recipes = ENGINES[engine]
    if quality not in recipes:
        pytest.skip("the frequency-domain engine serves HQ / VHQ only (by design)")
    plan = dev.Plan(in_rate, out_rate, quality)
    bits = plan.precision_bits
    frames = in_rate  # one second
    x, fp, fs, n0 = _probes(in_rate, out_rate, plan, frames, dtype)
    xt = torch.from_numpy(x[:, :, None].copy()).cuda()
    y = dev.resample_tensor(plan, xt, kernel=kernel).cpu().numpy()[:, :, 0].astype(np.float64)
    assert y.shape[1] == plan.out_len(frames)

    f64 = dtype == np.float64
    ripple = 2.0 ** -(bits - 1)
    amp_tol = ripple if f64 else max(ripple, F32_AMP_TOL)
    att_spec_db = (bits + 1) * 20 * np.log10(2)
    # float32 I/O rounds the OUTPUT to 2^-24 whatever the arithmetic: its floor is the float32 one
    floor_db = -(att_spec_db - 3.0) if f64 else max(-(att_spec_db - 3.0), F32_FLOOR_DB)
    rms_in = 1.0 / np.sqrt(2.0)

    # (i) pass band: amplitude; (ii) what is left beside the tone (images when up-sampling, aliases, noise)
    for i, f in enumerate(fp):
        amp, res = _fit(y[i], f, out_rate)
        assert abs(amp - 1.0) <= amp_tol, (engine, quality, f, amp - 1.0, amp_tol)
        assert 20 * np.log10(max(res, 1e-300) / rms_in) <= floor_db, (engine, quality, f, 20 * np.log10(res / rms_in), floor_db)
    # (ii) stop band (down-sampling): a tone above the output Nyquist must be gone
    for i, f in enumerate(fs):
        seg = y[N_PASS + i][int(0.2 * y.shape[1]):int(0.8 * y.shape[1])]
        lvl = 20 * np.log10(max(np.sqrt(np.mean(seg ** 2)), 1e-300) / rms_in)
        assert lvl <= floor_db, (engine, quality, f, lvl, floor_db)
    # (iii) impulse on a period boundary: symmetric response, peak at its own output instant
    h = y[-1]
    c = n0 * plan.L // plan.M
    half = int((plan.taps // 2 - 2) * plan.L / plan.M)
    assert int(np.argmax(np.abs(h))) == c
    left, right = h[c - half:c][::-1], h[c + 1:c + 1 + half]
    sym_tol = (1e-12 if kernel == EXACT else 2e-9) if f64 else 2e-7
    assert np.max(np.abs(left - right)) <= sym_tol * np.abs(h[c]), (engine, quality, float(np.max(np.abs(left - right)) / np.abs(h[c])))
    assert np.all(h[:c - half - 2 * plan.L] == 0) or kernel != EXACT   # (the exact engine's support is exact; the FFT engine leaks rounding noise)
