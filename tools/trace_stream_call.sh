#!/bin/bash
# HIP API / copy / kernel breakdown of ResampleStream.resample_chunk, int16 44.1k -> 16k VHQ mono: trace_stream_call.sh [chunk]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CH=${1:-96000}
cat > /tmp/strc.py <<PY
import sys, time; sys.path.insert(0, "$R/python-soxr_amd")
import numpy as np, soxr_amd as soxr
x = (np.random.default_rng(0).standard_normal(44100 * 120) * 5000).astype(np.int16)
rs = soxr.ResampleStream(44100, 16000, 1, dtype="int16", quality="VHQ")
t0 = time.perf_counter(); n = 0
for a in range(0, len(x), $CH):
    rs.resample_chunk(x[a:a + $CH]); n += 1
print("us per call", (time.perf_counter() - t0) / n * 1e6, "calls", n)
PY
rm -rf /tmp/trs; rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats -d /tmp/trs -o t -- python /tmp/strc.py > /tmp/trs.log 2>&1
tail -2 /tmp/trs.log
python - <<PY
import sqlite3, glob
con = sqlite3.connect(glob.glob("/tmp/trs/**/*.db", recursive=True)[0])
for r in con.execute("select * from top limit 14").fetchall():
    print("%-70s calls %6d  total %10.1f us  avg %9.2f us  %5.1f%%" % (str(r[0])[:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
