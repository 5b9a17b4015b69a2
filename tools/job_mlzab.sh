#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so
for v in "" "NL_MLZ_PERSIST=1" "NL_MLZ_PERSIST=1 NL_MLZ_WGS_PER_CU=2" "NL_MLZ_SPLIT=1" ""; do
  echo "== $v"
  env $v python tools/ab_stride.py d 2:512:4096 2:300:4096 2:256:4096 3:512:4096 2>&1 | grep -v amdgpu.ids
done
