cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/tests_gpu_serial.log 2>&1 ) 2>&1 | grep real; tail -3 gpurun_out/tests_gpu_serial.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
