#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded_frame_stride or lent_frames" 2>&1 | tail -5
(timeout 2400 python tools/ab_stride.py 0,d,32832,49216 \
  2:512:4096 2:300:4096 2:256:4096 2:200:4096 2:128:4096 2:64:4096 2:32:4096 2:24:4096 2:16:4096 2:8:4096 \
  3:16:4096 3:24:4096 3:32:4096 3:128:4096 3:512:4096 0:128:4096 1:128:4096 5:128:4096 4:32:4096 4:128:4096 \
  2:128:4096::w 3:128:4096::w 2:512:4096::w 2:32:512:4096 2:300:1024:4096 2:128:512:4096 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/pad3.txt
