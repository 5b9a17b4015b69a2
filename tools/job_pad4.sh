#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded_frame_stride or lent_frames" 2>&1 | tail -40
(
P1=0,1040,4160,8256,12352,16448,20544,24640,32832,41024,49216,65600,81984,98368,131136
timeout 1200 python tools/ab_stride.py $P1 0:128:4096 1:128:4096 2:128:4096 2:32:4096 4:128:4096 0:32:4096 1:32:4096
timeout 1200 python tools/ab_stride.py 0,8256,16448,24640,32832,49216 2:512:4096 2:300:4096
timeout 1200 python tools/ab_stride.py 0,41024,65600,81984,98368,131136 2:512:4096 2:300:4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pad4.txt
