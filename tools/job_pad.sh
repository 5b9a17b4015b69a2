#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
 for pad in 0 1088 4160 272; do
  export NL_STRIDE_PAD=$pad
  echo "== pad $pad"
  python tools/ab_flags.py 2 512 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
  python tools/ab_flags.py 2 128 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
  python tools/ab_flags.py 3 512 512 1536 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
  python tools/ab_flags.py 0 128 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
 done
done
