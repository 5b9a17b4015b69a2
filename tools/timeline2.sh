#!/bin/bash
# kernel timeline of the last passes of an ab_flags.py run (run on the GPU box):
#   tools/timeline2.sh <tag> <mode> <frames> <rows> <row0> <image_rows> <flags>   -> gpurun_out/timeline_<tag>.txt
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $repo/tools/ab_flags.py $1 $2 $3 $4 $5 1 $6 > /tmp/tl.out 2> /tmp/tl.log
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $out/timeline_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-24:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %7.1f  dur %8.1f  grid %8s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:60]))
    prev_end = max(prev_end, e)
PY
grep -v amdgpu.ids /tmp/tl.out | tail -2 >> $out/timeline_$tag.txt
echo "== $tag"; cat $out/timeline_$tag.txt
