#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch" 2>&1 | tail -3
for rep in 1 2; do for w in "2 128 512" "2 100 512" "2 80 512" "2 128 1024"; do python tools/wall_probe.py $w 8192 2>&1 | grep -v amdgpu; done; done
bash tools/timeline2.sh tile128_tf3 2 128 512 1536 4096 0 > /dev/null; tail -8 gpurun_out/timeline_tile128_tf3.txt | cut -c1-130
