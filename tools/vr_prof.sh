#!/bin/bash
# Kernel trace of the 96 000-frame variable-rate stream calls (tools/vr_big.py): which kernels a call runs and for how long.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/vrprof
rm -rf $OUT && mkdir -p $OUT
cd $R
python tools/vr_big.py > $OUT/plain.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python tools/vr_big.py > $OUT/trace.log 2>&1
cat $OUT/plain.log
find $OUT/t -name '*kernel_stats.csv' | head -1 | xargs cut -c1-200 | head -20
