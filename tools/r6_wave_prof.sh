#!/bin/bash
# round 6: per-wave stamps and PMC passes of the one-wave-per-pair kernel on the batch shard
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_wave_prof
rm -rf $OUT && mkdir -p $OUT
cd $R
{
bash tools/with_variant.sh trace python tools/trace_wave.py
B="python tools/wave_check.py 128 10"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES -d $OUT/p1 -o p -- $B > $OUT/p1.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/p2 -o p -- $B > $OUT/p2.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE -d $OUT/p3 -o p -- $B > $OUT/p3.log 2>&1
python tools/pmc_summary.py $OUT/p*/*.db 2>&1 | grep -v "k_tile\|per-grid.*k_fft_pair\|  void" 
} > $R/gpurun_out/r6_wave_prof.txt 2>&1
tail -80 $R/gpurun_out/r6_wave_prof.txt
