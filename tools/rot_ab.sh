cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for r in 0 4; do for v in nt nont; do
  echo -n "[$v ROTATE=$r] "
  if [ $v = nt ]; then ROTATE=$r python tools/run_workload.py batch 200 2>&1 | tail -n 1; else ROTATE=$r tools/with_variant.sh nont python tools/run_workload.py batch 200 2>&1 | tail -n 1; fi
done; done; done
