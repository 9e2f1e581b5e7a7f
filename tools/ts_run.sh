#!/bin/bash
# two-stage form: parity tests, then per-kernel times (rocprofv3 stats) of three ratios
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_two_stage.py -q -x 2>&1 | tail -2
python tools/two_stage_check.py 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
for cfg in "48000 44101 VHQ 2880000 2" "44101 48000 VHQ 2880000 2" "44100 16001 VHQ 2880000 2" "32000 96001 VHQ 1440000 2"; do
  rm -rf /tmp/tsp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/tsp -o t -- python $GRAFT_REPO_ROOT/tools/two_stage_prof.py $cfg 30 > /tmp/tsp.log 2>&1
  echo "== $cfg"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/tsp/*.db 2>/dev/null | grep -v "at::native\|rocclr\|per-grid" | grep "k_poly\|k_fft\|k_tile\|k_interp" | head -6
done
