#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/alloc_lottery.hip -o /tmp/alloc_lottery 2>/dev/null
timeout 600 /tmp/alloc_lottery 28 | tee gpurun_out/alloc_lottery28.txt
