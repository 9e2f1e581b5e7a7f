#!/bin/bash
# PMC passes over the two-stage job (k_poly): tools/prof_poly.sh [in out quality frames channels]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="${*:-48000 44101 VHQ 2880000 2} 12"
T="timeout -k 5 150"
$T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d /tmp/pp1 -o p -- python $R/tools/two_stage_prof.py $ARGS > /tmp/pp1.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR -d /tmp/pp2 -o p -- python $R/tools/two_stage_prof.py $ARGS > /tmp/pp2.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pp3 -o p -- python $R/tools/two_stage_prof.py $ARGS > /tmp/pp3.log 2>&1
cd $R; python tools/pmc_summary.py $(find /tmp/pp1 /tmp/pp2 /tmp/pp3 -name "*.db") 2>&1 | grep -A9 "^   (.k_poly" | cut -c1-120
