#!/bin/bash
# Developer probe -- run ON THE GPU BOX from the repo root:
#     tools/pmc_probe.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...] -- bench.py arguments
# One rocprofv3 --pmc pass (with --kernel-trace only) per counter group; prints per-kernel averages.
set -u
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for counters in "${groups[@]}"; do
    name=$(echo "$counters" | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $counters -d "$out/${tag}_probe_$name" -o p -- \
        python "$repo/bench.py" --steps 2 --warmup 1 --no-cpu --no-also "$@" > /dev/null 2> "$out/${tag}_probe_$name.log"
    db=$(find "$out/${tag}_probe_$name" -name 'p_results.db' | head -1)
    if [ -n "$db" ]; then python "$repo/tools/pmc_dump.py" "$db"; else tail -5 "$out/${tag}_probe_$name.log"; fi
done
find "$out" -name '*.db' -path "*${tag}_probe_*" -delete
