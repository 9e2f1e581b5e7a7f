#!/bin/bash
# whole GPU suite + smoke + the contract bench line on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5_suite_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r5_smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
tail -c 800 gpurun_out/r5_bench.err
cat gpurun_out/r5_suite_tests.txt; tail -2 gpurun_out/r5_smoke.txt
