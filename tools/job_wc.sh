#!/bin/bash
# round 5: winsorization cascade plans at 16 / 24 frames (NL_WCAS), A/B inside one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for plan in "" "1:8,1:12:4,2:16:2,0:0:1" "1:8,1:12:2,2:16:2,0:0:2" "1:8,1:12:4,2:16:4,0:0:1" "1:8,2:16:4,0:0:2" "1:8,2:16:4,0:0:1" "2:8,2:16:4,0:0:2" "1:12,2:16:4,0:0:2"; do
  for n in 16 24; do
    if [ -z "$plan" ]; then unset NL_WCAS; else export NL_WCAS="$plan"; fi
    echo -n "plan '${plan:-default}' : "
    python tools/ab_flags.py 3 $n 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/stack_sigma.*//'
  done
done
