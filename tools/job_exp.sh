#!/bin/bash
# experiment runner: A/B of library variants on one workload (tools/ab_flags.py arguments in $CFG)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/exp; mkdir -p $O; rm -f $O/ab.log
CFG=${CFG:-"2 512 4096 0 4096 3 0"}
for rep in 1 2; do
for lib in "" $LIBS; do
  if [ -z "$lib" ]; then unset NLSTACK_LIB; else export NLSTACK_LIB=$PWD/$lib; fi
  echo "== lib: ${lib:-default}" | tee -a $O/ab.log
  timeout 300 python tools/ab_flags.py $CFG 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
done
