#!/bin/bash
# PMC passes over k_interp_tile, two outputs per lane against one (debug build, HIPSOXR_NO_INTERP_PAIR=1): 48000 -> 44101 VHQ stereo 60 s float32
R=${GRAFT_REPO_ROOT:-/root/repo}
export HIPSOXR_LIBRARY=$R/python-soxr_amd/_variants/dbg/libhipsoxr.so
cd /tmp && export TMPDIR=/tmp
ARGS="48000 44101 VHQ 2880000 ${1:-2} 6 exact"
T="timeout -k 5 200"
for v in pair single; do
  if [ $v = single ]; then export HIPSOXR_NO_INTERP_PAIR=1; else unset HIPSOXR_NO_INTERP_PAIR; fi
  rm -rf /tmp/ip1 /tmp/ip2 /tmp/ip3
  $T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d /tmp/ip1 -o p -- python $R/tools/two_stage_prof.py $ARGS > /tmp/ip1.log 2>&1
  $T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d /tmp/ip2 -o p -- python $R/tools/two_stage_prof.py $ARGS > /tmp/ip2.log 2>&1
  $T rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES_EQ_64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d /tmp/ip3 -o p -- python $R/tools/two_stage_prof.py $ARGS > /tmp/ip3.log 2>&1
  echo "== [$v]"
  python $R/tools/pmc_summary.py $(find /tmp/ip1 /tmp/ip2 /tmp/ip3 -name "*.db") 2>&1 | grep -A9 "k_interp_tile" | grep -v "per-grid" | cut -c1-110
done
