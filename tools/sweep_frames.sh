#!/bin/bash
# pass / dominant-kernel time over the frame count, one line per (mode, frames): tools/sweep_frames.sh <rows> <mode> <n> [<n> ...]   (run on the GPU box)
rows=$1; mode=$2; shift; shift
for n in "$@"; do python tools/ab_flags.py $mode $n $rows 0 4096 1 0 2>&1 | grep -v amdgpu | cut -c1-200; done
