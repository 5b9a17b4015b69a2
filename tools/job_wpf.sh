#!/bin/bash
# round 5: weighted replays of 129 ... 512 frames with every frame of a work item's four pixels in registers (PF = 4 / 8)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wpf; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "weighted or large_stacks or wave_per_pixel or partition" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for pf in 1 0; do
  for a in "--weighted --frames 512 --height 1024" "--weighted --frames 256 --height 1024" "--weighted --frames 160 --height 1024" "--weighted --mode 3 --frames 512 --height 1024" "--weighted --mode 3 --frames 256 --height 1024"; do
    echo -n "NL_COOP_PF=$pf  "; NL_COOP_PF=$pf bash tools/qb.sh "--no-cpu $a"
  done
done | tee $O/ab.txt
