#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pad in 0 64 128 256 512 1024 2048 4096 4160 8192 8256 16384 16448 32832 65600 1048640 0; do
  export NL_STRIDE_PAD=$pad
  echo "== pad $pad"
  python tools/ab_flags.py 2 512 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
  python tools/ab_flags.py 2 128 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
  python tools/ab_flags.py 2 32 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
done
