#!/bin/bash
# round 5, after the padded frame stride: the whole profile set again (tools/profile_r05.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_r05.sh > gpurun_out/profile_r05b.log 2>&1
tail -5 gpurun_out/profile_r05b.log
