#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for plan in "" "1:6,1:12:4,0:0:4" "1:6,1:14:4,0:0:4" "1:6,2:14:4,0:0:4" "1:6,1:10:4,2:16:4,0:0:4" "1:5,1:10:4,2:16:4,0:0:4" "1:6,1:12:8,2:16:4,0:0:4" "1:6,1:12:4,2:16:8,0:0:2"; do
  if [ -z "$plan" ]; then unset NL_WCAS; else export NL_WCAS="$plan"; fi
  for n in 16 20 24 32; do
    echo -n "plan '${plan:-default}' : "
    python tools/ab_flags.py 3 $n 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//; s/(min [0-9.]*)//g'
  done
done
for plan in "" "2:10,2:16:8,3:24:4,0:0:4" "2:12,2:16:8,0:0:4" "1:8,2:16:8,3:24:4,0:0:4" "2:12,3:20:8,0:0:4"; do
  if [ -z "$plan" ]; then unset NL_WCAS; else export NL_WCAS="$plan"; fi
  for n in 48 64 96; do
    echo -n "deep plan '${plan:-default}' : "
    python tools/ab_flags.py 3 $n 4096 0 4096 2 0 2>&1 | grep -v amdgpu.ids | sed 's/clips.*//; s/(min [0-9.]*)//g'
  done
done
NL_FUZZ_MODES=3 NL_FUZZ_N=1,128 NL_FUZZ_WEIGHTED=0.05 timeout 1500 python tests/sweeps/fuzz_parity.py 40000 61 2>&1 | tail -1
timeout 1500 python tests/sweeps/fuzz_parity.py 20000 62 2>&1 | tail -1
