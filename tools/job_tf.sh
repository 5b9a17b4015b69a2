#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch" 2>&1 | tail -15
python tools/ab_flags.py 2 128 512 1536 4096 3 0,8192 2>&1 | grep -v amdgpu | cut -c1-150
python tools/ab_flags.py 2 128 4096 0 4096 3 0,8192 2>&1 | grep -v amdgpu | cut -c1-150
python tools/ab_flags.py 2 100 4096 0 4096 3 0,8192 2>&1 | grep -v amdgpu | cut -c1-150
python tools/wall_probe.py 2 128 512 8192 2>&1 | grep -v amdgpu
python tools/wall_probe.py 2 128 4096 8192 2>&1 | grep -v amdgpu
