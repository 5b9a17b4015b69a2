#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so
for prio in 0 1; do
for v in "0" "50,50" "60,40" "40,35,25" "34,33,33" "30,27,23,20" "0"; do
  echo "== NL_CHUNKS=$v prio=$prio"
  NL_CHUNK_PRIO=$prio NL_CHUNKS=$v python tools/ab_flags.py 3 512 512 1536 4096 3 0 2>&1 | grep -v amdgpu.ids | cut -c1-150
done; done
echo "== full image winsor 512, chunks"
for v in "0" "50,50" "30,27,23,20"; do NL_CHUNK_PRIO=0 NL_CHUNKS=$v python tools/ab_flags.py 3 512 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
