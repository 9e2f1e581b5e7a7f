# per-stage times of k_chain_resident's workgroup 0 (trace build), then the normal build's call times
HIPSOXR_EXTRA_FLAGS="-DHIPSOXR_RES_TRACE" bash python-soxr_amd/build.sh > /dev/null 2>&1
timeout 200 python tools/time_stream_call.py 2>&1 | grep -A1 "msg 2024" | grep -v "^--" | head -8
bash python-soxr_amd/build.sh > /dev/null 2>&1
for i in 1 2; do timeout 200 python tools/time_stream_call.py 2>&1 | grep "^flags"; done
