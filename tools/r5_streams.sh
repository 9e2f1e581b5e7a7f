#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_tensor_stream.py tests/test_gpu_device_client.py tests/test_gpu_fft.py -x -q -m gpu 2>&1 | tail -8
python - <<'PY'
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "python-soxr_amd")
import bench, json
d = bench.configs4_stream(10)
print(json.dumps({k: (round(v["us_per_call"], 2) if isinstance(v, dict) and "us_per_call" in v else v) for k, v in d.items() if k.startswith("device")}, indent=0))
PY
bash tools/r5_c2.sh 2>&1 | tail -32
