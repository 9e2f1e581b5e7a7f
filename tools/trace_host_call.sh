#!/bin/bash
# HIP API / copy / kernel breakdown of soxr.resample on a 60 s mono float32 host array
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/hostc1.py <<PY
import sys; sys.path.insert(0, "$R/python-soxr_amd")
import numpy as np, soxr_amd as soxr
x = (np.random.default_rng(0).standard_normal(48000 * 60) * 0.25).astype(np.float32)
for _ in range(25):
    soxr.resample(x, 48000, 44100, "VHQ")
PY
rm -rf /tmp/tr; rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats -d /tmp/tr -o t -- python /tmp/hostc1.py > /tmp/tr.log 2>&1
python - <<PY
import sqlite3, glob
con = sqlite3.connect(glob.glob("/tmp/tr/**/*.db", recursive=True)[0])
for r in con.execute("select * from top limit 12").fetchall():
    print("%-60s calls %6d  total %10.1f us  avg %9.2f us  %5.1f%%" % (str(r[0])[:60], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
