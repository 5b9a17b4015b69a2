#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline2.sh winsor16c 3 16 4096 0 4096 0 > /dev/null; tail -13 gpurun_out/timeline_winsor16c.txt
bash tools/timeline2.sh winsor24c 3 24 4096 0 4096 0 > /dev/null; tail -13 gpurun_out/timeline_winsor24c.txt
