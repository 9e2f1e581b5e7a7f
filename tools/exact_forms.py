#!/usr/bin/env python
"""Exact f32 engine (k_tile_mfma_p), every launch form forced in turn (HIPSOXR_DEBUG_TILE_FORM, debug-switch build) against
launch_tile's own choice, over job sizes: launch time and a digest of the result.  tools/exact_forms.py [child form frames clips]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-soxr_amd"))
    import torch
    from soxr_amd import device as dev
    out = {}
    for frames, clips in json.loads(sys.argv[2]):
        g = torch.Generator(device="cuda"); g.manual_seed(frames)
        x = torch.randn((clips, frames, 1), device="cuda", generator=g) * 0.25
        plan = dev.Plan(48000, 44100, "VHQ")
        y = dev.resample_tensor(plan, x, kernel=dev.KERNEL_EXACT)
        job = dev.PreparedJob(plan, x, y, kernel=dev.KERNEL_EXACT)
        for _ in range(5): job.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(40): job.launch()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 40)
        out["%dx%d" % (clips, frames)] = [best, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]]
    print("FORMS " + json.dumps(out))
    sys.exit(0)
# sizes in 64-period slabs (a period = 160 input frames): slabs * 64 * 160 frames
SIZES = [(150 * 10240 - 77, 1), (282 * 10240 - 1280, 1), (376 * 10240, 1), (260 * 10240, 2), (768 * 10240, 1), (1024 * 10240 + 333, 1), (47 * 10240, 30), (1500 * 10240, 1)]
dbg = os.path.join(ROOT, "python-soxr_amd", "_variants", "dbg", "libhipsoxr.so")
res = {}
for form in (0, 1, 2, 3, 4):
    env = dict(os.environ, HIPSOXR_LIBRARY=dbg)
    if form: env["HIPSOXR_DEBUG_TILE_FORM"] = str(form)
    r = subprocess.run([sys.executable, __file__, "child", json.dumps(SIZES)], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("FORMS ")]
    if not line:
        print("form", form, "failed:", r.stderr[-800:]); continue
    res[form] = json.loads(line[-1][6:])
names = {0: "chosen", 1: "64 whole", 2: "64 split", 3: "32 whole", 4: "32 split"}
for key in res[0]:
    t = {f: res[f][key][0] for f in res}
    same = len({res[f][key][1] for f in res}) == 1
    best = min(t[f] for f in t if f)
    print("%-16s %s  chosen/best %.3f  bit-identical %s" % (key, "  ".join("%s %.1f" % (names[f], t[f]) for f in sorted(t)), t[0] / best, same))
