#!/bin/bash
# The gate a library commit has to pass (ON THE GPU BOX, from the repo root; rule since round 6: no library commit after the
# last run of this script that ended "ok"):   tools/final_check.sh [out-dir]
#   1. the -m gpu suite through the C ABI   2. smoke()   3. the default bench line   4. tools/check_bench.py on it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/final}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q -n 4 > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log; tail -3 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python tools/check_bench.py $O/bench_default.json | tee $O/check_bench.txt
