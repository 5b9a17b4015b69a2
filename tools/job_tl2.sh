#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline2.sh sigma128 2 128 4096 0 4096 0 > /dev/null; tail -12 gpurun_out/timeline_sigma128.txt
bash tools/timeline2.sh sigma32 2 32 4096 0 4096 0 > /dev/null; tail -8 gpurun_out/timeline_sigma32.txt
bash tools/timeline2.sh winsor128 3 128 4096 0 4096 0 > /dev/null; tail -8 gpurun_out/timeline_winsor128.txt
