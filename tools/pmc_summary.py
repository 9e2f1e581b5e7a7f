#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel mean duration and mean PMC counters."""
import collections
import glob
import sqlite3
import sys


def short(name):
    """Kernel name without the hipsoxr:: noise but WITH its template arguments (float and double instances
    of one kernel share grid and block sizes: a 48-character prefix would merge their counters)."""
    return name.replace("hipsoxr::", "").replace("void ", "")[:150]


def main(paths):
    for db in paths:
        c = sqlite3.connect(db)
        print("==", db)
        try:
            for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                print("  %-60s calls %5d  avg %12.1f us  %5.1f%%" % (r[0][:60], r[1], r[3], r[4]))
        except sqlite3.Error as e:
            print("  (no top_kernels: %s)" % e)
        try:  # per (kernel, grid) launch durations: bench.py times two workloads with the same kernel
            q = ("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(end-start), min(end-start), "
                 "max(end-start) from kernels group by name, grid_x, grid_y, grid_z")
            for r in c.execute(q):
                if "hipsoxr" in r[0]:
                    print("  per-grid: %-100s grid (%d,%d,%d) block %d  n=%d  avg %.2f us  min %.2f  max %.2f" % (
                        short(r[0])[:100], r[1], r[2], r[3], r[4], r[5], r[6] / 1e3, r[7] / 1e3, r[8] / 1e3))
        except sqlite3.Error:
            pass
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        try:
            for r in c.execute("select kernel_name, counter_name, value, grid_size, workgroup_size from counters_collection"):
                if "hipsoxr" in r[0]:
                    agg[(short(r[0]), r[3], r[4])][r[1]].append(r[2])
        except sqlite3.Error:
            pass
        for k, v in agg.items():
            print("  ", k)
            for n, x in sorted(v.items()):
                print("      %-28s %16.1f  (n=%d)" % (n, sum(x) / len(x), len(x)))


if __name__ == "__main__":
    main(sys.argv[1:] or glob.glob("gpurun_out/prof/*/*.db"))
