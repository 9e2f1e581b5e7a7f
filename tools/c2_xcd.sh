cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  echo -n "[product] "; python tools/run_workload.py c2 200 2>&1 | tail -n 1
  echo -n "[xcdc] "; tools/with_variant.sh xcdc python tools/run_workload.py c2 200 2>&1 | tail -n 1
done
