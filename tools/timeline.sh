#!/bin/bash
# kernel timeline of a few passes (run on the GPU box): tools/timeline.sh <bench args>  -> gpurun_out/timeline.txt
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o t -- python $repo/bench.py --steps 4 --warmup 3 --no-cpu --no-also "$@" > /dev/null 2> /tmp/tl.log
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $out/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e)
PY
cat $out/timeline.txt
