"""The one-wave-per-block-pair kernel of the frequency-domain engine (csrc/fftwave.hip, round 6): 48k <-> 44.1k float32
unit-stride columns.  The product takes it only for large batches of the up direction (where it measures faster), so the
coverage comes from the debug-switch build with HIPSOXR_DEBUG_WAVE_MIN=1 — EVERY eligible job on the wave kernel, both
directions — against the oracle's float64 direct form on the oracle's own bank, at the engine's bar (<= 1e-6 relative RMS;
measured ~1.8e-7): lengths around a block's and a pair's kept run, planar batches, every 4-byte phase of input and output
columns, ragged batches, determinism.  The same probe with HIPSOXR_FFT_NO_WAVE (k_fft_pair2 everywhere) must stay in the
same class and differ from it (two kernels, two roundings) — or the switch would not be switching.  And the product's own
rule: a 128 x 10 s batch of the up direction lands on the wave kernel and agrees with the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DBG_LIB = os.path.join(os.path.dirname(HERE), "python-soxr_amd", "_variants", "dbg", "libhipsoxr.so")


def _probe(env_extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HIPSOXR_")}
    env.update(env_extra)
    assert os.path.exists(DBG_LIB), "build.sh makes the debug-switch build beside the product"
    env["HIPSOXR_LIBRARY"] = DBG_LIB
    r = subprocess.run([sys.executable, os.path.join(HERE, "_wave_probe.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("WAVE_PROBE ")][-1]
    return json.loads(line[len("WAVE_PROBE "):])


@pytest.fixture(scope="module")
def wave():
    return _probe({"HIPSOXR_DEBUG_WAVE_MIN": "1"})


@pytest.fixture(scope="module")
def pair2():
    return _probe({"HIPSOXR_FFT_NO_WAVE": "1"})


def test_wave_kernel_within_tolerance_of_oracle(wave):
    for k, v in wave.items():
        if k.endswith("_deterministic"):
            assert v is True, k
        elif not k.endswith("_sha"):
            assert v <= 1e-6, (k, v)


def test_wave_kernel_is_what_ran(wave, pair2):
    """The two kernels round differently: equal digests would mean the switch did nothing."""
    for d in ("down", "up"):
        assert wave[d + "_sha"] != pair2[d + "_sha"], d
        assert 0 < pair2[d + "_vs_exact"] <= 1e-6 and 0 < wave[d + "_vs_exact"] <= 1e-6
    for k, v in pair2.items():
        if not k.endswith(("_sha", "_deterministic")):
            assert v <= 1e-6, (k, v)


def test_product_rule_takes_the_wave_kernel_for_a_large_up_batch(oracle):
    """128 clips x 10 s, 44.1k -> 48k VHQ mono: 8832 block pairs >= the 8192 the product asks for.  AUTO's result equals the
    forced-FFT result bit for bit (one kernel), is not the exact engine's, and is within the bar of the oracle on sampled clips."""
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(44100, 48000, "VHQ")
    torch.manual_seed(7)
    x = torch.randn((128, 441000, 1), device="cuda") * 0.25
    y = dev.resample_tensor(plan, x)
    assert y.shape == (128, 480000, 1)
    assert torch.equal(y, dev.resample_tensor(plan, x, kernel=5))
    for c in (0, 77, 127):
        xc = x[c, :, 0].cpu().numpy()
        ref = oracle.resample(xc, 44100, 48000, "VHQ", mode="ref")
        yc = y[c, :, 0].cpu().numpy().astype(np.float64)
        rms = float(np.sqrt(np.mean(ref ** 2)))
        assert float(np.sqrt(np.mean((yc - ref) ** 2))) / rms <= 1e-6
        assert float(np.abs(yc - ref).max()) <= 4e-5 * rms
        assert not np.array_equal(y[c].cpu().numpy(), dev.resample_tensor(plan, x[c:c + 1], kernel=6)[0].cpu().numpy())
