#!/bin/bash
# round 5: winsorization cascade with travelling columns -- parity, then A/B (developer switch 8192 = gather again)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wc; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "winsor or sweep or kat or developer" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for n in 16 20 24 32; do
  python tools/ab_flags.py 3 $n 4096 0 4096 3 0,8192 2>&1 | grep -v amdgpu.ids | sed 's/stack_sigma.*//'
done
bash tools/timeline2.sh winsor16b 3 16 4096 0 4096 0 > /dev/null; tail -13 gpurun_out/timeline_winsor16b.txt
