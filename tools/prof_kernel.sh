#!/bin/bash
# PMC passes for one bench configuration (batch only by default). Usage: prof_kernel.sh <kernel-id>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-0}
OUT=/tmp/profk$K
rm -rf $OUT && mkdir -p $OUT
cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu --kernel $K"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/p1 -o p -- $B > $OUT/p1.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY -d $OUT/p2 -o p -- $B > $OUT/p2.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_IFETCH GRBM_GUI_ACTIVE -d $OUT/p3 -o p -- $B > $OUT/p3.log 2>&1
python tools/pmc_summary.py $OUT/p*/*.db > $OUT/summary.txt 2>&1
grep -v "at::native\|rocclr" $OUT/summary.txt | cut -c1-200 | tee $R/gpurun_out/profk$K.txt | head -80
