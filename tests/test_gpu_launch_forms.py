"""Every launch FORM of the canonical-order engine gives the same bits (round 3).

`launch_tile` (csrc/kernels.hip) now picks, per job, the slab size (64 / 32 / 16 periods), the unit split, and — for
the general-period kernel on small jobs — runs a row tile's two half-chains on two waves that meet through LDS.  None of
that may change a single output: an output is a pure function of its own taps, evaluated in the canonical order.  Sizes
below are chosen to land in each form (sizes in 64-period slabs: 10 and 520 -> 32-period slabs split, 47 -> 64-period
slabs split, 256 / 376 -> 32-period slabs whole, 768 -> 64-period slabs whole; general-period kernel: up to 96 slabs the
small-job form, beyond it 64-period slabs), and every result is compared with
  * the oracle's canonical-order port, bit for bit, on windows (first, last, random), and
  * the round-2 form of the same launch (HIPSOXR_DEBUG_SLAB64 / HIPSOXR_DEBUG_NO_HALVES in a child process), bit for bit
    over the WHOLE signal."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = 6

_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from soxr_amd import device as dev
a, b, q, dt, frames, ch, seed, out = float(sys.argv[2]), float(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
rng = np.random.default_rng(seed)
x = rng.standard_normal((frames, ch)) * 0.25
x = {"float32": x.astype(np.float32), "float64": x, "int16": (x * 20000).astype(np.int16), "int32": (x * 2 ** 30).astype(np.int32)}[dt]
xt = torch.from_numpy(x if ch > 1 else x[:, 0].copy()).cuda()
y = dev.resample_tensor(dev.Plan(a, b, q), xt, kernel=6)
np.save(out, y.cpu().numpy())
"""


def _run(tmp_path, name, env_extra, a, b, q, dt, frames, ch, seed):
    out = str(tmp_path / (name + ".npy"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("HIPSOXR_")}
    env.update(env_extra)
    env["HIPSOXR_LIBRARY"] = os.path.join(ROOT, "python-soxr_amd", "_variants", "dbg", "libhipsoxr.so")  # the build that reads HIPSOXR_DEBUG_*
    subprocess.run([sys.executable, "-c", _CHILD, os.path.join(ROOT, "python-soxr_amd"), str(a), str(b), q, dt, str(frames), str(ch), str(seed), out],
                   check=True, env=env, timeout=600)
    return np.load(out)


def _input(dt, frames, ch, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((frames, ch)) * 0.25
    return {"float32": x.astype(np.float32), "float64": x, "int16": (x * 20000).astype(np.int16), "int32": (x * 2 ** 30).astype(np.int32)}[dt]


@pytest.mark.parametrize("slabs,dt", [(10, "float32"), (47, "float32"), (256, "int16"), (376, "float32"), (520, "int16"), (768, "float32")])
def test_planar_kernel_forms(oracle, tmp_path, slabs, dt):
    frames = slabs * 64 * 160 - 37                                   # 48k -> 44.1k: a period is 160 input frames
    y = _run(tmp_path, "new", {}, 48000, 44100, "VHQ", dt, frames, 1, 5)
    y_old = _run(tmp_path, "old", {"HIPSOXR_DEBUG_SLAB64": "1"}, 48000, 44100, "VHQ", dt, frames, 1, 5)
    assert y.dtype == y_old.dtype and np.array_equal(y, y_old)
    x = _input(dt, frames, 1, 5)[:, 0]
    pl = oracle.plan(48000, 44100, "VHQ")
    rng = np.random.default_rng(slabs)
    mode = "port_f32"
    for k0 in [0, len(y) - 300] + [int(k) for k in rng.integers(0, len(y) - 300, 4)]:
        if dt == "float32":
            assert np.array_equal(y[k0:k0 + 300], oracle.resample_channel(pl, x, mode, k0=k0, n_out=300)), k0
    # (int16: rounding and saturation ride on the same accumulations — the whole-signal equality with the round-2 form
    #  above is the check; that form is what tests/test_gpu_parity.py holds against the oracle)


@pytest.mark.parametrize("a,b,dt,frames,ch", [(44100, 16000, "int16", 20000, 1), (44100, 16000, "float32", 400000, 1), (44100, 48000, "float32", 100000, 2),
                                               (44100, 16000, "int32", 96000, 1), (44100, 16000, "float64", 150000, 2), (44100, 16000, "float32", 3000000, 1)])
def test_general_period_kernel_forms(oracle, tmp_path, a, b, dt, frames, ch):
    y = _run(tmp_path, "new", {}, a, b, "VHQ", dt, frames, ch, 9)
    y_old = _run(tmp_path, "old", {"HIPSOXR_DEBUG_SLAB64": "1"}, a, b, "VHQ", dt, frames, ch, 9)
    y_one = _run(tmp_path, "one", {"HIPSOXR_DEBUG_NO_HALVES": "1"}, a, b, "VHQ", dt, frames, ch, 9)
    assert np.array_equal(y, y_old) and np.array_equal(y, y_one)
    if dt in ("float32", "float64"):
        x = _input(dt, frames, ch, 9)
        pl = oracle.plan(a, b, "VHQ")
        mode = "port_f32" if dt == "float32" else "port_f64"
        yc, xc = (y[:, ch - 1], x[:, ch - 1].copy()) if ch > 1 else (y, x[:, 0].copy())
        rng = np.random.default_rng(frames)
        for k0 in [0, len(yc) - 200] + [int(k) for k in rng.integers(0, len(yc) - 200, 3)]:
            assert np.array_equal(yc[k0:k0 + 200], oracle.resample_channel(pl, xc, mode, k0=k0, n_out=200)), k0


def test_chosen_form_is_near_the_best():
    """`launch_tile` picks slab size and unit split from a cost model fitted to one box's sweep; a clock or driver change could
    de-tune it silently.  For eight job sizes (150 .. 1500 slabs of 64 periods, one and several columns) every form is forced in
    turn (HIPSOXR_DEBUG_TILE_FORM, debug-switch build: tools/exact_forms.py): the rule's own choice must be within 10 % of the
    best forced form, and every form must give the same bits."""
    # boxes of the pool differ by up to 25 % in clock behaviour and a single sweep is noisy at the 10 % level: best of three
    # sweeps per size (a real de-tuning shows in all three), bit-identity in every one of them
    best = {}
    for attempt in range(3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exact_forms.py")], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = [l for l in r.stdout.splitlines() if "chosen/best" in l]
        assert len(rows) == 8, r.stdout[-2000:]
        for i, l in enumerate(rows):
            assert l.rstrip().endswith("bit-identical True"), l
            ratio = float(l.split("chosen/best")[1].split()[0])
            best[i] = min(best.get(i, 9.), ratio)
        if max(best.values()) <= 1.10:
            break
    assert max(best.values()) <= 1.10, best
