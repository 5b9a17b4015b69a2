#!/bin/bash
# round 5: guarded linear fit -- dynamic instruction counts (PMC) of the stages, guarded and bit-exact
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/lfg; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
for v in ${VS:-1 0}; do
  NL_LFG=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d $O/pm_$v -o p -- python $R/bench.py --steps 2 --warmup 1 --preheat-steps 0 --no-cpu --no-also --mode 5 --frames ${FR:-128} > /dev/null 2> $O/pm_$v.log
  python $R/tools/profile_summary.py pmc "$(find $O/pm_$v -name 'p_results.db' | head -1)" > $O/pmc_lfg$v.txt
  find $O/pm_$v -name '*.db' -delete
  cat $O/pmc_lfg$v.txt
done
