#!/bin/bash
# PMC passes over the batch workload only (bench.py, AUTO engine).  Usage: prof_batch.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-batch}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/p1 -o p -- $B > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p2 -o p -- $B > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_IFETCH -d $OUT/p3 -o p -- $B > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p4 -o p -- $B > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p5 -o p -- $B > $OUT/p5.log 2>&1
python tools/pmc_summary.py $OUT/trace/*.db $OUT/p*/*.db > $OUT/summary.txt 2>&1
grep -A12 "k_fft_pair" $OUT/summary.txt | grep -v "k_tile\|k_stream" | head -150
