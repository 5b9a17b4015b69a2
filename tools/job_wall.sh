#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/wall_probe.py 2 128 512 32 2>&1 | grep -v amdgpu
python tools/wall_probe.py 2 32 512 32 2>&1 | grep -v amdgpu
python tools/wall_probe.py 2 128 4096 32 2>&1 | grep -v amdgpu
python tools/wall_probe.py 2 32 4096 32 2>&1 | grep -v amdgpu
