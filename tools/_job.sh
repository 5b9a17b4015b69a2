cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/final_check.sh gpurun_out/final6a
( time NLSTACK_LIB=$GRAFT_REPO_ROOT/nightlight_amd/libnlstack_exp.so timeout 3000 python -m pytest tests -m gpu -x -q -n 4 > gpurun_out/final6a/tests_gpu_exp.log 2>&1 ) 2>&1 | grep real; tail -2 gpurun_out/final6a/tests_gpu_exp.log
