#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/frame_layout.hip -o /tmp/frame_layout 2>/dev/null
timeout 600 /tmp/frame_layout | tee gpurun_out/frame_layout.txt
