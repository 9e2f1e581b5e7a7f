"""Parity against a LIVE libsoxr, if one exists at run time.

libsoxr — the code the hot path actually lives in — is absent from the reference checkout (empty
submodule) and from this image, so these tests SKIP here and parity stays "unpinned" (DESIGN.md §2).
They are armed for the day a libsoxr appears (python-soxr in site-packages or a system libsoxr.so):
then the north-star bar is asserted against the real thing — 1e-6 relative RMS for float I/O on
band-limited input (where every spec-compliant design must agree; on white noise the transition
band, 0.9115-1.0 x Nyquist, differs by O(0.1) between any two designs), the reference's own
tolerances otherwise (tests/test_resample.py: 1e-4 on tones, +-2 LSB for integers).
"""
import numpy as np
import pytest

from oracle import live_libsoxr

LIVE = live_libsoxr.probe()
needs_live = pytest.mark.skipif(LIVE is None, reason="libsoxr: absent (no python-soxr package, no system libsoxr.so)")


def test_probe_never_resolves_to_this_repository():
    """The probe must not mistake our libsoxr-named ABI / `soxr` alias package for the reference."""
    assert LIVE is None or not LIVE.version.startswith("hipsoxr")


def _band_limited(n, frac, seed):
    from test_design_independent import band_limited_noise
    return band_limited_noise(n, frac, seed)


@needs_live
@pytest.mark.parametrize("in_rate,out_rate,quality", [(48000, 44100, "VHQ"), (48000, 44100, "HQ"), (44100, 16000, "VHQ")])
def test_oracle_vs_live_libsoxr_band_limited(oracle, in_rate, out_rate, quality):
    x = _band_limited(4 * in_rate, 0.85 * min(1.0, out_rate / in_rate), 21)
    want = LIVE.resample(x, in_rate, out_rate, quality)
    got = oracle.resample(x, in_rate, out_rate, quality, mode="ref")
    assert len(got) == len(want)
    err = np.sqrt(np.mean((got - want) ** 2)) / np.sqrt(np.mean(want ** 2))
    assert err <= 1e-6, err


@needs_live
@pytest.mark.gpu
@pytest.mark.parametrize("in_rate,out_rate,quality", [(48000, 44100, "VHQ"), (48000, 44100, "HQ"), (44100, 16000, "VHQ")])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_gpu_vs_live_libsoxr_band_limited(soxr, in_rate, out_rate, quality, dtype):
    x = _band_limited(4 * in_rate, 0.85 * min(1.0, out_rate / in_rate), 22).astype(dtype)
    want = LIVE.resample(x, in_rate, out_rate, quality).astype(np.float64)
    got = soxr.resample(x, in_rate, out_rate, quality=quality).astype(np.float64)
    assert len(got) == len(want)
    err = np.sqrt(np.mean((got - want) ** 2)) / np.sqrt(np.mean(want ** 2))
    assert err <= 1e-6, err


@needs_live
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["int16", "int32"])
def test_gpu_vs_live_libsoxr_integers(soxr, dtype):
    """+-2 LSB: libsoxr seeds its int16 dither randomly (reference tests/test_resample.py:208-211)."""
    x = (_band_limited(2 * 48000, 0.8, 23) * 4 * 5000).astype(dtype)
    want = LIVE.resample(x, 48000, 22050, "HQ").astype(np.int64)
    got = soxr.resample(x, 48000, 22050, quality="HQ").astype(np.int64)
    assert len(got) == len(want) and np.abs(got - want).max() <= 2


# ---- round 5: the first box that holds a libsoxr settles SURVEY.md §8c in ONE run -----------------------------------
# Every committed design-independent fixture, BASELINE configs[0] / configs[1] at full size, and the error split by band:
# two spec-compliant designs agree in the pass band to their ripple (the 1e-6 class) and may differ freely inside the
# transition band (0.9115-1.0 x the lower Nyquist for VHQ: SURVEY.md §7.3-1).  A failure here therefore NAMES the
# difference — "pass band off" is a design / gain / alignment mismatch, "transition band only" is window or cut-off
# placement — instead of reporting one RMS number.

def _band_errors(got, want, rate, lo_nyq, edges=(0.0, 0.5, 0.85, 0.9115, 1.0)):
    """Relative error energy of (got - want) per band of the OUTPUT spectrum, bands in fractions of the lower Nyquist."""
    n = min(len(got), len(want))
    w = np.hanning(n)
    E, W = np.abs(np.fft.rfft((got[:n] - want[:n]) * w)) ** 2, np.abs(np.fft.rfft(want[:n] * w)) ** 2
    f = np.fft.rfftfreq(n, 1.0 / rate) / lo_nyq
    out = {}
    for a, b in zip(edges[:-1], edges[1:]):
        m = (f >= a) & (f < b)
        out[f"{a:g}-{b:g}"] = float(np.sqrt(E[m].sum() / max(W[m].sum(), 1e-300)))
    out["above"] = float(np.sqrt(E[f >= 1.0].sum() / max(W.sum(), 1e-300)))
    return out


def _report(name, got, want, out_rate, lo_nyq):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    total = float(np.sqrt(np.mean((got - want) ** 2)) / np.sqrt(np.mean(want ** 2)))
    bands = _band_errors(got if got.ndim == 1 else got[:, 0], want if want.ndim == 1 else want[:, 0], out_rate, lo_nyq)
    print(f"[live libsoxr] {name}: rel rms {total:.3e}; per band (x lower Nyquist): " + ", ".join(f"{k}: {v:.2e}" for k, v in bands.items()))
    return total, bands


def _ref_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_ref_vectors.json")) as f:
        return json.load(f)["cases"]


@needs_live
@pytest.mark.parametrize("case", _ref_cases(), ids=lambda c: c["name"])
def test_every_design_independent_fixture_vs_live_libsoxr(case):
    """The committed fixtures (oracle float64 direct form on the oracle's own bank) against libsoxr on the same seeded
    white-noise input: pass band (below 0.85 x the lower Nyquist) at the 1e-6 bar, the rest reported."""
    x = (np.random.default_rng(case["seed"]).standard_normal((case["frames"], case["channels"])) * 0.25).astype(case["dtype"])
    want = np.asarray(LIVE.resample(x if case["channels"] > 1 else x[:, 0], case["in_rate"], case["out_rate"], case["quality"]), np.float64)
    want = want.reshape(len(want), -1)
    assert want.shape[0] == case["out_frames"]                      # libsoxr's own length rule
    idx, vals = np.asarray(case["index"]), np.asarray(case["values"])
    err_idx = float(np.sqrt(np.mean((want[idx] - vals) ** 2)) / case["rms"])
    print(f"[live libsoxr] {case['name']}: fixture samples vs libsoxr rel rms {err_idx:.3e} (white noise: includes the transition band)")
    # band split needs whole signals: recompute the oracle on this input
    from oracle import oracle as o
    saved, o.bank_provider = o.bank_provider, None
    try:
        got = o.resample(x.astype(np.float64), case["in_rate"], case["out_rate"], case["quality"], mode="ref")
    finally:
        o.bank_provider = saved
    lo = min(case["in_rate"], case["out_rate"]) / 2
    _, bands = _report(case["name"], got, want, case["out_rate"], lo)
    assert bands["0-0.5"] <= 1e-6 and bands["0.5-0.85"] <= 1e-6, bands


@needs_live
@pytest.mark.gpu
@pytest.mark.parametrize("name,seconds,quality", [("configs0_10s_HQ", 10, "HQ"), ("configs1_60s_VHQ", 60, "VHQ")])
def test_baseline_configs_full_size_vs_live_libsoxr(soxr, name, seconds, quality):
    """BASELINE configs[0] (10 s HQ, the reference's README case) and configs[1] (60 s VHQ) at FULL size, mono float32
    48k -> 44.1k: the host surface (exact engine) and the AUTO device job (frequency-domain engine) against libsoxr —
    on the reference's own log sweep (tests/bench.py:32-36, band-limited below 23.9 kHz... the last octave enters the
    transition band) and on band-limited noise, where the north-star bar (1e-6 relative RMS) is asserted."""
    import torch
    from soxr_amd import device as dev
    n = 48000 * seconds
    t = np.arange(n) / 48000.0
    sweep = np.sin(2 * np.pi * 100.0 * seconds / np.log(239.0) * (np.exp(t / seconds * np.log(239.0)) - 1.0)).astype(np.float32)
    noise = _band_limited(n, 0.85 * 44100 / 48000, 31).astype(np.float32)
    plan = dev.Plan(48000, 44100, quality)
    for label, x, bar in (("log sweep 100 Hz -> 23.9 kHz", sweep, None), ("band-limited noise", noise, 1e-6)):
        want = np.asarray(LIVE.resample(x, 48000, 44100, quality), np.float64)
        host = soxr.resample(x, 48000, 44100, quality=quality)
        auto = dev.resample_tensor(plan, torch.from_numpy(x).cuda()).cpu().numpy()
        assert len(host) == len(want) == len(auto)
        for eng, got in (("host/exact", host), ("device/AUTO", auto)):
            total, bands = _report(f"{name} {label} {eng}", got, want, 44100, 22050)
            assert bands["0-0.5"] <= 1e-6 and bands["0.5-0.85"] <= 1e-6, (eng, label, bands)
            if bar is not None:
                assert total <= bar, (eng, label, total)
