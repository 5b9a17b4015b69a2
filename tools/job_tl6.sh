#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline2.sh tile128_tf 2 128 512 1536 4096 0 > /dev/null; tail -12 gpurun_out/timeline_tile128_tf.txt
bash tools/timeline2.sh tile128_tf32 2 128 512 1536 4096 32 > /dev/null; tail -10 gpurun_out/timeline_tile128_tf32.txt
