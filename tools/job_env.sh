#!/bin/bash
# experiment runner: A/B of environment settings ($ENVS: ';'-separated VAR=value lists, "-" = none) over workloads ($CFGS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/exp; mkdir -p $O; rm -f $O/env.log
IFS=';' read -ra CF <<< "${CFGS:-2 128 4096 0 4096 3 0}"
IFS=';' read -ra EV <<< "${ENVS:--}"
for cfg in "${CF[@]}"; do
  for rep in 1 2; do
    for ev in "${EV[@]}"; do
      echo -n "[$ev] " | tee -a $O/env.log
      if [ "$ev" = "-" ]; then timeout 300 python tools/ab_flags.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $O/env.log
      else env $ev timeout 300 python tools/ab_flags.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $O/env.log; fi
    done
  done
done
