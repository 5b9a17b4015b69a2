#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 1200 python tools/soak.py 2>&1 | tail -3
for f in 1 0 1 0; do
echo "== NL_EV_FENCE=$f"
for w in "2 128 512" "2 32 512" "2 128 4096" "2 32 4096" "3 16 4096" "2 512 4096" "3 512 512"; do
NL_EV_FENCE=$f python tools/wall_probe.py $w 0 2>&1 | grep -v amdgpu | head -1
done; done | tee gpurun_out/ev2/wall.txt
