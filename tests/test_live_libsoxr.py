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
