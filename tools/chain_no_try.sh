# experiments on k_chain / k_chain_resident staging (trace builds print per-stage times of workgroup 0)
for fl in "-DHIPSOXR_RES_TRACE" "-DHIPSOXR_RES_TRACE -DHIPSOXR_RES_TRACE_SPAN"; do
HIPSOXR_EXTRA_FLAGS="$fl" bash python-soxr_amd/build.sh > /dev/null 2>&1
for no in ${NOS:-8}; do echo "== $fl NO $no"; HIPSOXR_DEBUG_NO=$no timeout 200 python tools/time_stream_call.py 2>&1 | grep -A1 "msg 2024\|^flags" | grep -v "^--" | head -14; done; done
