cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -x -q -n 4 > $O/tests_gpu.log 2>&1 ) 2>&1 | grep real; tail -4 $O/tests_gpu.log
python tools/linfit_probe.py 128 > $O/linfit_probe.txt 2>&1; python tools/linfit_probe.py 64 >> $O/linfit_probe.txt 2>&1; python tools/linfit_probe.py 32 >> $O/linfit_probe.txt 2>&1; cat $O/linfit_probe.txt
tools/timeline.sh --mode 5 > /dev/null 2>&1; tail -12 gpurun_out/timeline.txt
