#!/bin/bash
# round 5: the cap on the winsorization loops behind the dominant kernel, re-tuned with the certificate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cap in 0 16 24 32 60 100; do
  if [ "$cap" = "0" ]; then unset NL_GEN_ROUND_CAP; else export NL_GEN_ROUND_CAP=$cap; fi
  for n in 16 24 32 64; do
    echo -n "cap ${cap} : "
    python tools/ab_flags.py 3 $n 4096 0 4096 2 32768 2>&1 | grep -v amdgpu.ids | sed 's/stack_sigma.*//; s/(min [0-9.]*)//g'
  done
done
