#!/bin/bash
# rocprofv3 passes over bench.py: kernel trace + stats, then PMC passes (counters in their own runs: gpurun refuses --pmc
# combined with other trace domains).  The traffic record (and the per-kernel average duration of the trace pass) is made
# by tools/make_traffic.py from trace / pmc1 / pmc3 / pmc4.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $R
ARGS="${BENCH_ARGS:---steps 20 --warmup 3 --kernels-only --windows 6 --sustained-s 0.5}"
SHORT="--steps 3 --warmup 1 --kernels-only --windows 2 --no-sustained"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p -- python bench.py $SHORT > $OUT/pmc1.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python bench.py $SHORT > $OUT/pmc3.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python bench.py $SHORT > $OUT/pmc4.log 2>&1
python tools/pmc_summary.py $OUT/trace/*.db $OUT/pmc*/*.db > $OUT/summary.txt 2>&1
grep -h '"metric"' $OUT/bench_trace.log > $OUT/bench_line.json
python tools/make_traffic.py $OUT $OUT/traffic.json > $OUT/traffic.log 2>&1
# the rocpd databases are tens of MB each; gpurun copies back at most 64 MiB: keep the summaries only
find $OUT -name "*.db" -delete
grep -v "at::native\|rocclr\|k_chain" $OUT/summary.txt | cut -c1-220 | head -150
cat $OUT/traffic.json
