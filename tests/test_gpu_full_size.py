"""(Exactness tests pin `kernel=EXACT`; the default AUTO engine for large float32 device jobs is the
frequency-domain one, which is compared with the exact result at the 1e-6 RMS bar.)

BASELINE.json's full-size configurations on the device-resident path, checked through
size-independent properties (the oracle would take minutes at these sizes) plus spot windows that
ARE compared with the oracle bit for bit:
  * shift invariance: delaying the input by M samples delays the output by exactly L samples, bit
    for bit (every output is a pure function of its own taps);
  * channel / clip independence: a multichannel or batched launch equals per-column launches;
  * linearity within float rounding;
  * random output windows equal the oracle's canonical-order port exactly.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXACT = 6   # hipsoxr_kernel_t: AUTO restricted to the canonical-order kernels


def _rel_rms(a, b):
    import torch
    return float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())


def _windows_match_oracle(oracle, plan_args, x_np, y_np, rng, n_windows=6, width=300):
    pl = oracle.plan(*plan_args)
    for k0 in [0, len(y_np) - width] + list(rng.integers(0, len(y_np) - width, n_windows)):
        k0 = int(k0)
        want = oracle.resample_channel(pl, x_np, "port_f32", k0=k0, n_out=width)
        assert np.array_equal(y_np[k0:k0 + width], want), f"window at {k0}"


def _windows_close_to_reference(oracle, plan_args, x_np, y_np, rng, n_windows=6, width=2000, tol=1e-6):
    """The frequency-domain engine's output against the oracle's float64 direct form, window by window (first and
    last outputs included): the north star's bar holds locally, not just as an average over the signal."""
    pl = oracle.plan(*plan_args)
    for k0 in [0, len(y_np) - width] + list(rng.integers(0, len(y_np) - width, n_windows)):
        k0 = int(k0)
        want = oracle.resample_channel(pl, x_np.astype(np.float64), "ref", k0=k0, n_out=width)
        err = y_np[k0:k0 + width].astype(np.float64) - want
        assert np.sqrt(np.mean(err ** 2)) <= tol * max(np.sqrt(np.mean(want ** 2)), 0.05), f"window at {k0}"


def test_config1_vhq_60s_mono(oracle):
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(48000, 44100, "VHQ")
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(2880000, device="cuda", generator=g) * 0.25
    y = dev.resample_tensor(plan, x, kernel=EXACT)
    assert y.shape[0] == 2646000
    y_auto = dev.resample_tensor(plan, x)                      # AUTO -> frequency-domain engine
    assert y_auto.shape == y.shape and _rel_rms(y_auto, y) <= 1e-6
    _windows_close_to_reference(oracle, (48000, 44100, "VHQ"), x.cpu().numpy(), y_auto.cpu().numpy(), np.random.default_rng(10))
    # shift invariance, exact
    xs = torch.cat([torch.zeros(plan.M * 5, device="cuda"), x])
    ys = dev.resample_tensor(plan, xs, kernel=EXACT)
    assert torch.equal(ys[plan.L * 5 + 2000:plan.L * 5 + 2000 + 2600000], y[2000:2602000])
    # both kernels, exact
    assert torch.equal(y, dev.resample_tensor(plan, x, kernel=1))
    # linearity
    x2 = torch.randn(2880000, device="cuda", generator=g) * 0.25
    lin = dev.resample_tensor(plan, 0.5 * x - 2.0 * x2, kernel=EXACT)
    comb = 0.5 * y - 2.0 * dev.resample_tensor(plan, x2, kernel=EXACT)
    assert (lin - comb).abs().max().item() < 2e-6
    # oracle windows, exact
    _windows_match_oracle(oracle, (48000, 44100, "VHQ"), x.cpu().numpy(), y.cpu().numpy(),
                          np.random.default_rng(0))


def test_config2_vhq_8ch_44k1_16k(oracle):
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(44100, 16000, "VHQ")
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((2646000, 8), device="cuda", generator=g) * 0.25
    y = dev.resample_tensor(plan, x, kernel=EXACT)
    assert tuple(y.shape) == (960000, 8)
    y_auto = dev.resample_tensor(plan, x)
    assert _rel_rms(y_auto, y) <= 1e-6
    for c in (0, 6):
        _windows_close_to_reference(oracle, (44100, 16000, "VHQ"), x[:, c].cpu().numpy(), y_auto[:, c].cpu().numpy(),
                                    np.random.default_rng(11 + c), n_windows=2, width=600)
    for c in (0, 5, 7):   # channel independence: interleaved launch == planar mono launch
        assert torch.equal(y[:, c], dev.resample_tensor(plan, x[:, c].contiguous(), kernel=EXACT))
    assert torch.equal(y, dev.resample_tensor(plan, x, kernel=1))
    _windows_match_oracle(oracle, (44100, 16000, "VHQ"), x[:, 3].cpu().numpy(), y[:, 3].cpu().numpy(),
                          np.random.default_rng(1), n_windows=3, width=100)


def test_config3_batch_of_clips(oracle):
    """One GPU's shard of the 1024-clip batch (128 x 10 s): batched launch == per-clip launches."""
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(48000, 44100, "VHQ")
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn((128, 480000, 1), device="cuda", generator=g) * 0.25
    y = dev.resample_tensor(plan, x, kernel=EXACT)
    assert tuple(y.shape) == (128, 441000, 1)
    y_auto = dev.resample_tensor(plan, x)
    assert _rel_rms(y_auto, y) <= 1e-6
    for clip in (0, 77, 127):
        _windows_close_to_reference(oracle, (48000, 44100, "VHQ"), x[clip, :, 0].cpu().numpy(), y_auto[clip, :, 0].cpu().numpy(),
                                    np.random.default_rng(20 + clip), n_windows=2)
    for clip in (0, 63, 127):
        assert torch.equal(y[clip, :, 0], dev.resample_tensor(plan, x[clip, :, 0].contiguous(), kernel=EXACT))
    _windows_match_oracle(oracle, (48000, 44100, "VHQ"), x[77, :, 0].cpu().numpy(), y[77, :, 0].cpu().numpy(),
                          np.random.default_rng(2), n_windows=3)


@pytest.mark.parametrize("chunk", [441, 4410, 96000])
@pytest.mark.parametrize("channels", [1, 2])
def test_config4_int16_stream(soxr, oracle, chunk, channels):
    rng = np.random.default_rng(5)
    n = 2646000 if chunk >= 4410 else 441000      # 60 s (10 s for the smallest chunk: 1000 launches)
    x = (rng.standard_normal((n, channels)) * 5000).astype(np.int16)
    rs = soxr.ResampleStream(44100, 16000, channels, dtype="int16", quality="VHQ")
    parts = [rs.resample_chunk(x[i:i + chunk], last=(i + chunk >= n)) for i in range(0, n, chunk)]
    y = np.concatenate(parts)
    assert y.shape[0] == n * 160 // 441
    one = soxr.resample(x, 44100, 16000, quality="VHQ")
    assert np.array_equal(y, one)                  # state carried across launches changes nothing
    pl = oracle.plan(44100, 16000, "VHQ")
    for k0 in (0, 123456 % (len(y) - 200), len(y) - 200):
        v = oracle.resample_channel(pl, x[:, 0].astype(np.float32), "port_f32", k0=k0, n_out=200)
        q, _ = oracle.quantize(v, np.int16, channel=0, k0=k0, dither=True, seed=0)
        assert np.array_equal(y[k0:k0 + 200, 0], q)


def test_prepared_job_equals_resample_tensor():
    """device.PreparedJob (what bench.py times: one C call per step) launches the same job."""
    import torch
    from soxr_amd import device as dev
    plan = dev.Plan(48000, 44100, "VHQ")
    x = torch.randn(300000, device="cuda") * 0.25
    y1 = dev.resample_tensor(plan, x)
    y2 = torch.empty_like(y1)
    job = dev.PreparedJob(plan, x, y2)
    job.launch()
    job.launch()
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    xb = torch.randn((3, 50000, 2), device="cuda") * 0.25
    yb = dev.resample_tensor(plan, xb, kernel=dev.KERNEL_EXACT)
    yb2 = torch.empty_like(yb)
    dev.PreparedJob(plan, xb, yb2, kernel=dev.KERNEL_EXACT).launch()
    torch.cuda.synchronize()
    assert torch.equal(yb, yb2)


def test_bench_two_ranks_harness():
    """bench.py's N > 1 path (env ranks, bank broadcast, barrier-bracketed timing, max over ranks, one
    JSON line from rank 0), exercised with two ranks sharing this box's GPU over gloo
    (BENCH_DIST_BACKEND=gloo is a test harness switch; real runs use RCCL, one rank per GPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "3", "--no-batch"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert d["cpu_baseline"] is None and "roofline" in d
