#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB_REPS=2 python tools/ab_stride.py d,d,d,d,d,d,21824,21824,21824,21824,21824,21824,0,0,0,0 0:128:4096 1:128:4096 2:128:4096 2>&1 | grep -v amdgpu.ids
