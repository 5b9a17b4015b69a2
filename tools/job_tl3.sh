#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline2.sh tile128 2 128 512 1536 4096 0 > /dev/null; tail -14 gpurun_out/timeline_tile128.txt
bash tools/timeline2.sh tile32 2 32 512 1536 4096 0 > /dev/null; tail -10 gpurun_out/timeline_tile32.txt
bash tools/timeline2.sh c3 3 512 512 1536 4096 0 > /dev/null; tail -10 gpurun_out/timeline_c3.txt
