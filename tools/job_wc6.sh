#!/bin/bash
# round 5: early tail of the winsorization cascade -- parity, A/B (developer switch 32768 = tail behind the cascade), timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wc; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "winsor or sweep or kat or developer or extreme or goal" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for n in 16 24 32 48 64; do
  python tools/ab_flags.py 3 $n 4096 0 4096 3 0,32768,49152 2>&1 | grep -v amdgpu.ids | sed 's/stack_sigma.*//'
done
bash tools/timeline2.sh winsor16d 3 16 4096 0 4096 0 > /dev/null; tail -14 gpurun_out/timeline_winsor16d.txt
