#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline2.sh tile128_f2 2 128 512 1536 4096 2 > /dev/null; tail -12 gpurun_out/timeline_tile128_f2.txt
bash tools/timeline2.sh tile128_f32 2 128 512 1536 4096 32 > /dev/null; tail -12 gpurun_out/timeline_tile128_f32.txt
bash tools/timeline2.sh tile128_f34 2 128 512 1536 4096 34 > /dev/null; tail -12 gpurun_out/timeline_tile128_f34.txt
