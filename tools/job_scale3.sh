#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_capi_exports.py -m gpu -x -q 2>&1 | tail -3
bash tools/job_scale.sh 2>&1 | grep -E "^rows|^frames"
