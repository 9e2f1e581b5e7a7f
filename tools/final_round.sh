#!/bin/bash
# what a round commits under profiles/: the rocprofv3 summary + traffic record (tools/prof_bench.sh), then — with that record
# in place — the full default bench line, and the tail of the GPU test suite.   tools/final_round.sh <tag, e.g. r05>
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
mkdir -p gpurun_out
bash tools/prof_bench.sh > gpurun_out/prof_stdout.txt 2>&1
cp gpurun_out/prof/traffic.json profiles/${TAG}_traffic.json
cp gpurun_out/prof/summary.txt gpurun_out/${TAG}_rocprofv3_summary.txt
cp gpurun_out/prof/bench_line.json gpurun_out/${TAG}_bench_line_under_trace.json
cp gpurun_out/prof/traffic.json gpurun_out/${TAG}_traffic.json
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -c 300 gpurun_out/${TAG}_bench.err
