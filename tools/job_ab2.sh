#!/bin/bash
# A/B of two library builds, interleaved: tools/job_ab2.sh <variant .so>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1
for rep in 1 2 3; do
 for lib in "" $V; do
  if [ -z "$lib" ]; then unset NLSTACK_LIB; else export NLSTACK_LIB=$PWD/$lib; fi
  echo "== rep $rep lib ${lib:-default}"
  for a in "2 128" "2 64" "2 32" "0 128" "0 64" "3 128" "4 128"; do set -- $a; python tools/ab_flags.py $1 $2 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'; done
 done
done
