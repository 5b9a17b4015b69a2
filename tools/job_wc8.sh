#!/bin/bash
# round 5: the certificate as committed -- whole GPU suite, then A/B per frame count (developer switch 16384 = off)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wc; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; echo "rc=$?" >> $O/tests_all.log
tail -3 $O/tests_all.log
for n in 12 15 16 17 20 24 25 32 40 48 64 80 96 100 112 128; do
  python tools/ab_flags.py 3 $n 4096 0 4096 2 0,16384 2>&1 | grep -v amdgpu.ids | sed 's/stack_sigma.*//; s/(min [0-9.]*)//g'
done | tee $O/cert_ab.txt
