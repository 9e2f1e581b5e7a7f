#!/bin/bash
# rocprofv3 passes over bench.py: kernel trace + stats, then PMC passes (counters in their own runs).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $R
ARGS="${BENCH_ARGS:---steps 20 --warmup 3 --no-cpu --windows 5}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc1 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu --windows 2 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu --windows 2 > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu --windows 2 > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu --windows 2 > $OUT/pmc4.log 2>&1
python tools/pmc_summary.py $OUT/trace/*.db $OUT/pmc*/*.db > $OUT/summary.txt 2>&1
grep -h '"metric"' $OUT/bench_trace.log > $OUT/bench_line.json
cat $OUT/summary.txt
