#!/bin/bash
# A/B timing of library builds: tools/ab_lib.sh [nightlight_amd/libnlstack_<variant>.so ...]  (run on the GPU box)
for lib in "" "$@"; do
  echo "== lib: ${lib:-default}"
  if [ -z "$lib" ]; then unset NLSTACK_LIB; else export NLSTACK_LIB=$PWD/$lib; fi
  tools/qb.sh "--no-cpu" "--no-cpu --mode 3" "--no-cpu --frames 512" "--no-cpu --mode 3 --frames 512 --height 512 --image-height 4096" "--no-cpu --mode 4" "--no-cpu --mode 0" "--no-cpu --mode 5"
done
