#!/bin/bash
# round 5: certificate parameters x cascade plans at 16 / 24 frames
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cert in "3,2" "2,2" "2,1" "3,1" "4,2" "3,3" "2,3"; do
 for plan in "" "1:6,1:12:4,2:16:4,0:0:4" "1:10,1:12:4,2:16:4,0:0:4" "1:8,2:16:4,0:0:4" "1:8,1:12:4,0:0:4"; do
  export NL_WCERT="$cert"
  if [ -z "$plan" ]; then unset NL_WCAS; else export NL_WCAS="$plan"; fi
  for n in 16 24; do
    echo -n "cert $cert plan '${plan:-default}' : "
    python tools/ab_flags.py 3 $n 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//'
  done
 done
done
