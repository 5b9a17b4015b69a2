#!/bin/bash
# Collect the artefacts committed under profiles/ -- run ON THE GPU BOX from the repo root:
#     tools/gpu_profile.sh <tag> [bench.py arguments...]
# e.g.  tools/gpu_profile.sh sigma128            (default workload: 128 x 4096^2 sigma clip)
#       tools/gpu_profile.sh sigma512 --frames 512
# Writes into gpurun_out/:
#     <tag>_bench.json          the bench line of the profiled command (rocprofv3 attached)
#     <tag>_kernel_stats.txt    rocprofv3 --kernel-trace --stats summary
#     <tag>_pmc.txt             FETCH_SIZE / WRITE_SIZE / busy counters, each from its own --pmc pass
#     <tag>_traffic.json        HBM bytes per launch of the dominant kernel (guide's gfx950 correction)
# PMC passes use --kernel-trace only (never together with sys/hip/hsa traces).
set -u
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
steps="--steps 20 --warmup 5 --no-cpu --no-also"      # enough passes for the clocks to settle

rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats" -o p -- \
    python "$repo/bench.py" $steps "$@" > "$out/${tag}_bench.json" 2> "$out/${tag}_stats.log"
python "$repo/tools/profile_summary.py" stats "$(find "$out/${tag}_stats" -name 'p_results.db' | head -1)" \
    > "$out/${tag}_kernel_stats.txt"

: > "$out/${tag}_pmc.txt"
for counters in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
    name=$(echo "$counters" | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $counters -d "$out/${tag}_pmc_$name" -o p -- \
        python "$repo/bench.py" --steps 3 --warmup 1 --preheat-steps 0 --no-cpu --no-also "$@" > /dev/null 2> "$out/${tag}_pmc_$name.log"
    db=$(find "$out/${tag}_pmc_$name" -name 'p_results.db' | head -1)
    if [ -n "$db" ]; then python "$repo/tools/profile_summary.py" pmc "$db" >> "$out/${tag}_pmc.txt"; fi
done
python "$repo/tools/profile_summary.py" traffic "$out/${tag}_pmc.txt" "$out/${tag}_bench.json" > "$out/${tag}_traffic.json"
# the databases are large; only the summaries travel back
find "$out" -name '*.db' -path "*${tag}_*" -delete
cat "$out/${tag}_kernel_stats.txt"
cat "$out/${tag}_traffic.json"
