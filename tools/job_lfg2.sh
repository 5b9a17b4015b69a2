#!/bin/bash
# round 5: guarded linear fit -- per-kernel times (rocprofv3 kernel trace) of the 128-frame pass, guarded and bit-exact
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/lfg; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "linear or linfit or mode_matches or ties_and or infinite or newton or c4" > $O/tests_lf.log 2>&1; echo "rc=$?" >> $O/tests_lf.log
tail -3 $O/tests_lf.log
cd /tmp
for v in 1 0; do
  NL_LFG=$v rocprofv3 --kernel-trace --stats -d $O/st_$v -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu --no-also --mode 5 --frames ${FR:-128} > $O/bench_lfg$v.json 2> $O/st_$v.log
  python $R/tools/profile_summary.py stats "$(find $O/st_$v -name 'p_results.db' | head -1)" > $O/kernel_stats_lfg$v.txt
  find $O/st_$v -name '*.db' -delete
  head -12 $O/kernel_stats_lfg$v.txt
  python3 -c "
import json
for l in open('$O/bench_lfg$v.json'):
    if l.startswith('{'):
        d = json.loads(l); print('NL_LFG=$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])
"
done
