#!/usr/bin/env python
"""profiles/<name>_traffic.json from the rocprofv3 PMC passes of tools/prof_bench.sh (gpurun_out/prof/pmc{1,3,4}):
HBM traffic per launch (FETCH_SIZE x 2 — the gfx950 correction of MI355X_MICROARCH.md §HBM for wide coalesced reads —
plus WRITE_SIZE, both in KiB) and VALU wave-instructions per launch for the three bench workloads, keyed by the SHA-256
of the kernel sources they were collected on (bench.py `measured_counters` refuses a record whose hash has gone stale).
    python tools/make_traffic.py gpurun_out/prof profiles/r05_traffic.json"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_sources_sha16, KERNEL_SOURCES)

WORKLOADS = {  # name -> (kernel-name substring, grid predicate, algorithmic bytes)
    "configs1": ("k_fft_pair2<hipsoxr::PairSpec<2560, 2352", lambda gx, gy: gy == 1, 4 * (2880000 + 2646000)),
    "batch_shard": ("k_fft_pair2<hipsoxr::PairSpec<5120, 4704", lambda gx, gy: gy == 128, 4 * 128 * (480000 + 441000)),
    "batch_up": ("k_fft_wave<hipsoxr::WaveSpec<3528, 3840", lambda gx, gy: True, 4 * 128 * (441000 + 480000)),
    "configs2": ("k_fft_strided2<hipsoxr::PairSpec<4410, 1600", lambda gx, gy: True, 4 * 8 * (2646000 + 960000)),
    "float64": ("double, double>", lambda gx, gy: True, 8 * (2880000 + 2646000)),
    "arith_f64": ("double, float>", lambda gx, gy: True, 4 * (2880000 + 2646000)),
    "exact_engine": ("k_tile_mfma_p<float>", lambda gx, gy: True, 4 * (2880000 + 2646000)),
    # the two kernels of the arbitrary-ratio job (48000 -> 44101 stereo 60 s): bytes each kernel has to move
    "two_stage_poly": ("k_poly2<16, 0, true>", lambda gx, gy: True, 4 * 2 * (2880000 + 2 * 2646060)),
    "two_stage_fft": ("k_fft_strided2<hipsoxr::PairSpec<4096, 2048", lambda gx, gy: True, 4 * 2 * (2 * 2646060 + 2646060)),
}


def counters(db):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    # counters_collection has grid_size (threads) only; the kernels table has grid_x/y per dispatch: join on dispatch id
    try:
        rows = c.execute("select kernel_name, counter_name, value, grid_size, workgroup_size from counters_collection").fetchall()
    except sqlite3.Error:
        return agg
    for name, cn, val, gs, ws in rows:
        agg[(name, gs, ws)][cn].append(val)
    return agg


def pick(agg, sub, counter, want_threads=None):
    vals = []
    for (name, gs, ws), d in agg.items():
        if sub in name and counter in d and (want_threads is None or gs == want_threads):
            vals.extend(d[counter])
    return sum(vals) / len(vals) if vals else None


def trace_avg_us(prof_dir):
    """kernel name -> (grid threads -> average duration in us) from the kernel-trace pass (the figure bench.py prints as
    roofline.rocprof_avg_us beside its own HIP-event time)."""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        c = sqlite3.connect(_db(prof_dir, "trace"))
        rows = c.execute("select name, grid_x * grid_y * grid_z, (end - start) from kernels").fetchall()
    except (sqlite3.Error, SystemExit):
        try:
            rows = c.execute("select kernel_name, grid_size, (end_timestamp - start_timestamp) from kernel_dispatch").fetchall()  # (other rocpd schema)
        except Exception:
            return out
    for name, gs, ns in rows:
        out[name][gs].append(ns / 1e3)
    return out


def pick_us(tr, sub, want_threads=None):
    vals = []
    for name, d in tr.items():
        if sub in name:
            for gs, v in d.items():
                if want_threads is None or gs == want_threads:
                    vals.extend(v)
    vals = vals[len(vals) // 10:]            # (the first launches run at a ramping clock)
    return round(sum(vals) / len(vals), 3) if vals else None


def main(prof_dir, out_path):
    p1, p3, p4 = counters(_db(prof_dir, "pmc1")), counters(_db(prof_dir, "pmc3")), counters(_db(prof_dir, "pmc4"))
    tr = trace_avg_us(prof_dir)
    threads = {"configs1": None, "batch_shard": 128 * 50 * 384, "configs2": None, "float64": None, "arith_f64": None,
               "exact_engine": 1704 * 256}   # (the 60 s clip's launches only: bench.py also runs the batch on this kernel)
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ_INSTS_VALU passes (separate runs) of bench.py (tools/prof_bench.sh); "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE uncorrected; KiB",
           "kernel_sources": list(bench.KERNEL_SOURCES), "kernel_sources_sha16": bench.kernel_sources_sha16(), "workloads": {}}
    for wl, (sub, _, algo) in WORKLOADS.items():
        f, w, v = pick(p3, sub, "FETCH_SIZE", threads.get(wl)), pick(p4, sub, "WRITE_SIZE", threads.get(wl)), pick(p1, sub, "SQ_INSTS_VALU", threads.get(wl))
        if f is None or w is None:
            continue
        traffic = int((2 * f + w) * 1024)
        rec["workloads"][wl] = {"kernel": sub, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes": traffic, "algorithmic_bytes": algo,
                                "ratio": round(traffic / algo, 4), "valu_wave_insts": v, "rocprof_avg_us": pick_us(tr, sub, threads.get(wl))}
    with open(out_path, "w") as fo:
        json.dump(rec, fo, indent=1)
    print(json.dumps(rec, indent=1))


def _db(prof_dir, sub):
    d = os.path.join(prof_dir, sub)
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(".db"):
                return os.path.join(root, f)
    raise SystemExit(f"no .db under {d}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof"),
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", bench.TRAFFIC_FILE))
