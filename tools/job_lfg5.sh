#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" nightlight_amd/libnlstack_ch4.so; do
  if [ -z "$lib" ]; then unset NLSTACK_LIB; else export NLSTACK_LIB=$PWD/$lib; fi
  echo "== lib ${lib:-default}"
  for n in 128 64 32; do NL_LFG=0 python tools/ab_flags.py 5 $n 4096 0 4096 3 0 2>&1 | grep -v amdgpu.ids; done
done
