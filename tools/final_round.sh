#!/bin/bash
# what a round commits under profiles/: the rocprofv3 summary + traffic record (tools/prof_bench.sh), then — with that record
# in place — the full default bench line, and the tail of the GPU test suite.   tools/final_round.sh <tag, e.g. r04>
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
bash tools/prof_bench.sh > gpurun_out/prof_stdout.txt 2>&1
cp gpurun_out/prof/traffic.json profiles/${TAG}_traffic.json
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
