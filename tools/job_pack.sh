#!/bin/bash
# round 4: packed rounds phase + staged replay rows -- parity, then A/B (flags 128 = no staging; r3 library = round-3 kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pack; mkdir -p $O; rm -f $O/ab.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "not fullsize" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for lib in "" nightlight_amd/libnlstack_r3mlz.so; do
  if [ -z "$lib" ]; then unset NLSTACK_LIB; F=0,128; else export NLSTACK_LIB=$PWD/$lib; F=0; fi
  echo "== lib: ${lib:-default}" | tee -a $O/ab.log
  for cfg in "2 512 4096 0 4096" "2 300 2048 0 4096" "2 256 4096 0 4096" "3 512 512 1536 4096" "3 300 1024 0 4096" "3 200 2048 0 4096"; do
    timeout 300 python tools/ab_flags.py $cfg 3 $F 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
  done
done
