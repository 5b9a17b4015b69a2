#!/bin/bash
# round 5: linear fit -- parity tests, then timing of the bit-exact cascade (NL_LFG=0) and the guarded one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lfg; mkdir -p $O
NL_LFG=${LFG:-0} timeout 900 python -m pytest tests -m gpu -x -q -k "linear or linfit or mode_matches or ties_and or infinite or newton or c4" > $O/tests_lf.log 2>&1; echo "rc=$?" >> $O/tests_lf.log
tail -4 $O/tests_lf.log
for n in ${FRS:-128 64 32}; do
  timeout 600 python tools/ab_flags.py 5 $n 4096 0 4096 3 0,4096 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
