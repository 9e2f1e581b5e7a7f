"""GPU: the two-stage form of arbitrary ratios (csrc/twostage.hip — FFT engine at 1:2 / 2:1 + a short interpolated
polyphase stage in LDS) against the oracle's float64 direct form ON ITS OWN BANK: random integer and float rate pairs in
the style of the reference's tests/test_random.py:21-25 (seeded here), float32 and float64, mono and interleaved stereo, at
size (0.4 - 3 M frames), white noise — the composite filter is the plan's own prototype, so no band-limiting is needed.
Bars: float32 <= 1e-6 relative RMS (the engine's class), float64 <= 2e-9 — on windows that include the first and the last
outputs (the intermediate signal runs past both ends of the job: no output is patched by another engine)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs(seed, n):
    r = random.Random(seed)
    out = []
    while len(out) < n:
        a, b = (r.randint(8000, 96000), r.randint(8000, 96000)) if len(out) % 2 == 0 else (r.uniform(8000, 96000), r.uniform(8000, 96000))
        if 0.26 < b / a < 12:        # (the two-stage form serves down-sampling to 4:1; steeper ratios stay on the exact engine)
            out.append((a, b))
    return out


PAIRS = [(48000, 44101), (44101, 48000), (44100, 16001)] + _pairs(4, 5)


def _rms(v):
    return float(np.sqrt(np.mean(np.square(v, dtype=np.float64))))


@pytest.mark.parametrize("in_rate,out_rate", PAIRS)
@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-6), (np.float64, 2e-9)])
def test_two_stage_matches_the_oracle_direct_form(oracle, in_rate, out_rate, dtype, tol):
    import torch
    from soxr_amd import device as dev
    rng = np.random.default_rng(int(in_rate) % 1000 + 7)
    ch = 2 if int(in_rate) % 2 else 1
    frames = 400_000 if dtype == np.float64 else 1_200_000
    x = (rng.standard_normal((frames, ch)) * 0.25).astype(dtype)
    plan = dev.Plan(in_rate, out_rate, "VHQ")
    assert plan.phases > 0                                   # an interpolated-phase plan: no exact bank for this ratio
    xt = torch.from_numpy(x if ch > 1 else x[:, 0].copy()).cuda()
    y = dev.resample_tensor(plan, xt).cpu().numpy().reshape(-1, ch)
    ye = dev.resample_tensor(plan, xt, kernel=dev.KERNEL_EXACT).cpu().numpy().reshape(-1, ch)
    assert y.shape == ye.shape == (plan.out_len(frames), ch)
    if np.array_equal(y, ye):                                 # (a long polyphase table in float64 does not fit LDS: exact engine)
        assert dtype == np.float64 and in_rate / out_rate > 2.5, "AUTO did not take the two-stage form"
        return
    # the very ends (partial sums over the signal's first / last samples), against the exact engine: absolute, at the scale of the signal
    for sl in (slice(0, 64), slice(-64, None)):
        assert np.max(np.abs(y[sl].astype(np.float64) - ye[sl])) <= 8 * tol * 0.25, sl
    pl = oracle.plan(in_rate, out_rate, "VHQ")
    c = ch - 1
    # the oracle's float64 direct form on windows (its cost is 300-700 taps x a cubic per output): head, tail, three inside
    n_out = y.shape[0]
    for k0 in (0, n_out - 3000, n_out // 3, n_out // 2 + 777, (2 * n_out) // 3):
        ref = oracle.resample_channel(pl, x[:, c].astype(np.float64), "ref", k0=k0, n_out=3000)
        got = y[k0:k0 + 3000, c].astype(np.float64)
        assert _rms(got - ref) <= tol * max(_rms(ref), 0.05), (k0, _rms(got - ref) / max(_rms(ref), 0.05))
    # and the whole signal against the exact engine (same prototype, canonical order)
    assert _rms(y.astype(np.float64) - ye) <= tol * _rms(ye), _rms(y.astype(np.float64) - ye) / _rms(ye)


def test_two_stage_is_not_taken_where_it_does_not_apply():
    """Steep down-sampling (> 4:1), the 16-bit recipes, small jobs, integer I/O and KERNEL_EXACT stay on the canonical-order
    engine: AUTO == EXACT bit for bit."""
    import torch
    from soxr_amd import device as dev
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    x = torch.randn(300000, device="cuda", generator=g) * 0.25
    for a, b, q, t in ((96000, 8001, "VHQ", x), (48000, 44101, "MQ", x), (48000, 44101, "VHQ", x[:3000].contiguous()),
                       (48000, 44101, "VHQ", (x * 20000).to(torch.int16))):
        plan = dev.Plan(a, b, q)
        assert torch.equal(dev.resample_tensor(plan, t), dev.resample_tensor(plan, t, kernel=dev.KERNEL_EXACT)), (a, b, q)


def test_two_stage_batches_and_layouts(oracle):
    """Several clips x channels in one job, interleaved and planar, float32: every column within 1e-6 of the exact engine."""
    import torch
    from soxr_amd import device as dev
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    for ch, a, b in ((4, 44100, 48001), (3, 44100, 48001), (8, 48000, 44101), (2, 32000, 11025.5), (6, 22050, 48000.5)):
        plan = dev.Plan(a, b, "HQ")
        x = torch.randn((3, 200000, ch), device="cuda", generator=g) * 0.25     # [clips, frames, channels] interleaved
        for t in (x, x.permute(0, 2, 1).contiguous().permute(0, 2, 1)):        # ... and the same values planar
            y = dev.resample_tensor(plan, t)
            ye = dev.resample_tensor(plan, t, kernel=dev.KERNEL_EXACT)
            d = (y.double() - ye.double())
            rel = (d.pow(2).mean(dim=1).sqrt() / ye.double().pow(2).mean(dim=1).sqrt()).max().item()
            assert 0 < rel <= 1e-6, (ch, a, b, rel)


def test_two_stage_channel_pairs_on_views():
    """Round 5: interleaved channel pairs run on k_poly2 (8-byte frames, the table reads and the cubic per tap shared by the
    two channels) when both ends of the polyphase stage are channel-interleaved, frame strides are even and the data is
    8-byte aligned; everything else stays on the one-channel-per-pass kernel.  Views that are eligible (two channels of a
    four-channel tensor at an even offset), views that are not (odd channel offset: 4-byte aligned only; odd frame stride),
    results written into a view — each within 1e-6 of the exact engine on a contiguous copy, up- and down-sampling."""
    import torch
    from soxr_amd import device as dev
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x4 = torch.randn((260000, 4), device="cuda", generator=g) * 0.25
    x3 = torch.randn((260000, 3), device="cuda", generator=g) * 0.25
    for a, b in ((48000, 44101), (44101, 48000), (44100, 16001.5), (16000, 44099)):
        plan = dev.Plan(a, b, "VHQ")
        for view in (x4[:, 0:2], x4[:, 1:3], x4[:, 2:4], x3[:, 0:2], x3[:, 1:3], x4):
            ye = dev.resample_tensor(plan, view.contiguous(), kernel=dev.KERNEL_EXACT).double()
            y = dev.resample_tensor(plan, view)
            assert y.shape == ye.shape
            rel = float((y.double() - ye).pow(2).mean().sqrt() / ye.pow(2).mean().sqrt())
            assert 0 < rel <= 1e-6, (a, b, tuple(view.stride()), view.storage_offset(), rel)
            # ... and into two channels of a wider result (even and odd channel offsets)
            for off in (0, 1):
                wide = torch.zeros((y.shape[0], view.shape[1] + 2), device="cuda")
                dev.resample_tensor(plan, view, out=wide[:, off:off + view.shape[1]])
                assert torch.equal(wide[:, off:off + view.shape[1]], y) or \
                    float((wide[:, off:off + view.shape[1]].double() - ye).pow(2).mean().sqrt() / ye.pow(2).mean().sqrt()) <= 1e-6
                assert float(wide[:, :off].abs().sum()) == 0 and float(wide[:, off + view.shape[1]:].abs().sum()) == 0


def test_two_stage_split_columns():
    """Round 5: a column that is not half of an interleaved pair (mono, planar, odd channel counts) is split a whole number
    of phase periods in and its two segments run as the pair of k_poly2 (output k + h Ls has output k's fraction).  Lengths
    from below the threshold (single segment) through 1.7 ... 9 periods, the second segment ending inside a tile, mono /
    planar stereo / 3 interleaved channels, both directions: within 1e-6 of the exact engine overall AND pointwise at the
    float32 class over the whole signal (a seam or a tail in the wrong place would be O(1))."""
    import torch
    from soxr_amd import device as dev
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    for a, b in ((48000, 44101), (44101, 48000), (44100, 16001), (32000, 44099)):
        plan = dev.Plan(a, b, "VHQ")
        for frames in (8192, 30000, 41000, 48000, 60001, 100003, 144000, 200017, 400009):
            for shape in ("mono", "planar2", "inter3"):
                if shape != "mono" and frames not in (41000, 100003, 400009):
                    continue
                x = torch.randn((frames, {"mono": 1, "planar2": 2, "inter3": 3}[shape]), device="cuda", generator=g) * 0.25
                t = x[:, 0].contiguous() if shape == "mono" else x.t().contiguous().t() if shape == "planar2" else x
                y = dev.resample_tensor(plan, t).double()
                ye = dev.resample_tensor(plan, t, kernel=dev.KERNEL_EXACT).double()
                assert y.shape == ye.shape
                scale = float(ye.pow(2).mean().sqrt())
                rel = float((y - ye).pow(2).mean().sqrt()) / scale
                worst = float((y - ye).abs().max()) / scale
                assert rel <= 1e-6 and worst <= 1e-5, (a, b, frames, shape, rel, worst)


def _edge_pairs():
    r = random.Random(77)
    fixed = [(48000, 48001), (48001, 48000), (44100, 44100.5), (8000, 31999), (8000, 32001), (96000, 24001), (96000, 23999.5),
             (48000, 95999), (48000, 96001), (44100, 22051), (44100, 22049), (11025, 44101), (32000, 12345.678), (12345.678, 32000),
             (44100, 48000.01), (48000.01, 44100), (22050, 88201.5), (50000, 20001), (16000, 16001), (37800, 44056.5)]
    rnd = []
    while len(rnd) < 28:
        a, b = r.uniform(8000, 96000), r.uniform(8000, 96000)
        if 0.26 < b / a < 12:
            rnd.append((a, b) if len(rnd) % 3 else (int(a), int(b) | 1))
    return fixed + rnd


@pytest.mark.parametrize("quality", ["VHQ", "HQ"])
def test_two_stage_over_many_ratios(quality):
    """48 rate pairs — ratios a hair off 1, 1/2, 2, 4, 1/4, float rates, near-coprime integers — stereo float32, 0.3 M frames:
    AUTO (two-stage wherever the plan is interpolated) within 1e-6 relative RMS of the canonical-order engine on every pair,
    the first and last 256 outputs included at the same absolute scale."""
    import torch
    from soxr_amd import device as dev
    g = torch.Generator(device="cuda"); g.manual_seed(31)
    x = torch.randn((300000, 2), device="cuda", generator=g) * 0.25
    worst = 0.0
    for a, b in _edge_pairs():
        plan = dev.Plan(a, b, quality)
        y = dev.resample_tensor(plan, x)
        ye = dev.resample_tensor(plan, x, kernel=dev.KERNEL_EXACT)
        assert y.shape == ye.shape, (a, b)
        d = (y.double() - ye.double())
        scale = float(ye.double().pow(2).mean().sqrt())
        rel = float(d.pow(2).mean().sqrt()) / scale
        ends = max(float(d[:256].abs().max()), float(d[-256:].abs().max())) / scale
        assert rel <= 1e-6 and ends <= 8e-6, (a, b, quality, plan.phases, rel, ends)
        worst = max(worst, rel)
    assert worst > 0        # (at least one pair took the two-stage form)


def test_an_installed_bank_is_honoured_by_every_engine():
    """Round-4 advisor finding: the two-stage form samples the plan's ANALYTIC prototype, so a bank installed from outside
    (`hipsoxr_plan_set_bank`, the bank broadcast) must take the plan off that form — else AUTO float device jobs of
    >= 8192 frames would silently ignore it while every other path follows it.  A scaled bank scales the output of
    every engine; the plan's own bank re-installed (the broadcast's usual case) changes nothing and keeps the fast form."""
    import torch
    from soxr_amd import device as dev
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.randn(200000, device="cuda", generator=g) * 0.25
    plan = dev.Plan(48000, 44101.5, "VHQ")
    assert plan.phases > 0
    y_auto = dev.resample_tensor(plan, x).double()
    y_exact = dev.resample_tensor(plan, x, kernel=dev.KERNEL_EXACT).double()
    assert not torch.equal(y_auto.float(), y_exact.float())            # AUTO really is the two-stage form here ...
    assert float((y_auto - y_exact).norm() / y_exact.norm()) <= 1e-6   # ... and agrees with the exact engine
    same = dev.Plan(48000, 44101.5, "VHQ")
    same.set_bank(plan.bank())                                          # identical bank: a no-op, the fast form stays
    assert torch.equal(dev.resample_tensor(same, x), dev.resample_tensor(plan, x))
    half = dev.Plan(48000, 44101.5, "VHQ")
    half.set_bank(0.5 * plan.bank())
    for kernel in (dev.KERNEL_AUTO, dev.KERNEL_EXACT):
        got = dev.resample_tensor(half, x, kernel=kernel).double()
        assert float((got - 0.5 * y_exact).norm() / y_exact.norm()) <= 1e-6, kernel
    zero = dev.Plan(48000, 44101.5, "VHQ")
    zero.set_bank(np.zeros_like(plan.bank()))
    assert not dev.resample_tensor(zero, x).any()
    # round 6 (advisor): the flag is two-way — the DESIGNED bank installed again puts the plan back on the fast form, bit for bit
    half.set_bank(plan.bank())
    assert torch.equal(dev.resample_tensor(half, x), dev.resample_tensor(plan, x))
    half.set_bank(0.25 * plan.bank())                                   # ... and off it again
    got = dev.resample_tensor(half, x).double()
    assert float((got - 0.25 * y_exact).norm() / y_exact.norm()) <= 1e-6
