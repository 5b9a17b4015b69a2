#!/bin/bash
# round 5: invariant-interval certificate of the winsorization loops -- parity, then A/B (developer switch 16384 = off)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wc; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "winsor or sweep or kat or developer or extreme or goal" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for n in 16 24 32 64 96 128; do
  python tools/ab_flags.py 3 $n 4096 0 4096 3 0,16384 2>&1 | grep -v amdgpu.ids | sed 's/stack_sigma.*//'
done
