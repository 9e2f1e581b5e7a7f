#!/bin/bash
# A/B of variant builds on one box with power/clock: tools/var_ab.sh "<variant> [ENV=..]" ...   ("-" = product build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
W=${WORKLOAD:-batch}
CFGS=("$@")
for rep in 1 2; do
for cfg in "${CFGS[@]}"; do
  v=${cfg%% *}; envs=""; [ "$v" != "$cfg" ] && envs=${cfg#* }
  echo -n "[$v $envs] "
  if [ "$v" = "-" ]; then env $envs python tools/power_probe.py $W 2.0 2>&1 | tail -n 2 | tr '\n' ' '; else env $envs tools/with_variant.sh $v python tools/power_probe.py $W 2.0 2>&1 | tail -n 2 | tr '\n' ' '; fi
  echo
done; done
