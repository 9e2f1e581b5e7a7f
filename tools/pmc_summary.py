#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel mean duration and mean PMC counters."""
import collections
import glob
import sqlite3
import sys


def main(paths):
    for db in paths:
        c = sqlite3.connect(db)
        print("==", db)
        try:
            for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                print("  %-60s calls %5d  avg %12.1f us  %5.1f%%" % (r[0][:60], r[1], r[3], r[4]))
        except sqlite3.Error as e:
            print("  (no top_kernels: %s)" % e)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        try:
            for r in c.execute("select kernel_name, counter_name, value, grid_size, workgroup_size from counters_collection"):
                agg[(r[0][:48], r[3], r[4])][r[1]].append(r[2])
        except sqlite3.Error:
            pass
        for k, v in agg.items():
            if "hipsoxr" not in k[0]:
                continue
            print("  ", k)
            for n, x in sorted(v.items()):
                print("      %-28s %16.1f  (n=%d)" % (n, sum(x) / len(x), len(x)))


if __name__ == "__main__":
    main(sys.argv[1:] or glob.glob("gpurun_out/prof/*/*.db"))
