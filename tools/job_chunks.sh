#!/bin/bash
# round 4: chunked passes -- parity with the chunks forced on every fast-path pass, then A/B timing of plans
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/chunks; mkdir -p $O
NL_CHUNKS=40,30,20,10 timeout 900 python -m pytest tests -m gpu -x -q -k "not fullsize" > $O/tests_forced.log 2>&1; echo "forced rc=$?" >> $O/tests_forced.log
tail -3 $O/tests_forced.log
for plan in "0" "50,30,15,5" "34,30,20,10,4,2" "30,25,20,12,7,4,2" "60,30,10" "25,25,20,15,10,5"; do
  echo "== NL_CHUNKS=$plan" | tee -a $O/ab512.log
  NL_CHUNKS=$plan timeout 300 python tools/ab_flags.py 2 512 4096 0 4096 3 0,64 2>&1 | grep -v amdgpu.ids | tee -a $O/ab512.log
done
echo "== prio 0, NL_CHUNKS=34,30,20,10,4,2" | tee -a $O/ab512.log
NL_CHUNK_PRIO=0 NL_CHUNKS=34,30,20,10,4,2 timeout 300 python tools/ab_flags.py 2 512 4096 0 4096 3 0,64 2>&1 | grep -v amdgpu.ids | tee -a $O/ab512.log
