#!/bin/bash
# experiment runner: A/B of library variants ($LIBS) over several workloads (tools/ab_flags.py argument lists, ';'-separated in $CFGS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/exp; mkdir -p $O; rm -f $O/ab2.log
IFS=';' read -ra CF <<< "${CFGS:-2 128 4096 0 4096 3 0;2 32 4096 0 4096 3 0;2 512 4096 0 4096 2 0;3 128 4096 0 4096 3 0;3 512 512 1536 4096 3 0;3 16 4096 0 4096 3 0}"
for cfg in "${CF[@]}"; do
  for rep in 1 2; do
    for lib in "" $LIBS; do
      if [ -z "$lib" ]; then unset NLSTACK_LIB; else export NLSTACK_LIB=$PWD/$lib; fi
      echo -n "[${lib:-default}] " | tee -a $O/ab2.log
      timeout 300 python tools/ab_flags.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $O/ab2.log
    done
  done
done
