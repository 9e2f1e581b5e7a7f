#!/bin/bash
# Stall attribution of ONE workload's kernel from PMC passes (round 3; the thread-trace decoder library that
# `rocprofv3 --att` needs is not in this image — the attempt and its message are recorded first).
#   tools/prof_attr.sh <batch|clip|c2> [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-batch}; TAG=${2:-$W}
OUT=$R/gpurun_out/attr_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
B="python tools/run_workload.py $W 10"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
( timeout 120 rocprofv3 --att --kernel-trace -d $OUT/att -o a -- python tools/run_workload.py $W 2 ) > $OUT/att_attempt.log 2>&1; echo "att rc=$?" >> $OUT/att_attempt.log
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
pass() { n=$1; shift; timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o p -- $B > $OUT/$n.log 2>&1 || tail -3 $OUT/$n.log; }
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
pass p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS
pass p3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass p4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
# (a pass with the TA_* counters aborts inside rocprofv3 on this image and then hangs in its signal handler: left out)
pass p6 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass p7 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass p8 FETCH_SIZE
pass p9 WRITE_SIZE
python tools/pmc_summary.py $OUT/trace/*.db $OUT/p*/*.db > $OUT/summary.txt 2>&1
grep -v "at::native\|rocclr" $OUT/summary.txt | head -120
tail -5 $OUT/att_attempt.log
