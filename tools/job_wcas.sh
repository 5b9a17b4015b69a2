#!/bin/bash
# round 4: winsorization cascade -- parity, then A/B (flags 128 = no cascade) and stage plans
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wcas; mkdir -p $O; rm -f $O/ab.log
timeout 1500 python -m pytest tests -m gpu -x -q -k "not fullsize" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for n in 16 20 24 32 64 128; do
  timeout 300 python tools/ab_flags.py 3 $n 4096 0 4096 3 0,128 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
for plan in "1:6,1:8:3,1:12:3,0:0:4" "1:8,1:12:4,2:16:4,0:0:4" "2:12,1:16:8,0:0:4" "2:10,2:14:6,0:0:4" "2:12,2:16:8,3:24:4,0:0:4" "1:6,1:6:2,1:8:2,1:12:3,2:16:4,0:0:4"; do
  echo "== NL_WCAS=$plan" | tee -a $O/ab.log
  for n in 16 24 64 128; do
    NL_WCAS=$plan timeout 300 python tools/ab_flags.py 3 $n 4096 0 4096 3 0 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
  done
done
