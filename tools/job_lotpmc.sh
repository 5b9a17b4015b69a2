#!/bin/bash
# per-dispatch PMC counters of the allocation microbenchmark (8 allocations, the same streaming kernel): what differs between a
# fast and a slow allocation?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/alloc_lottery.hip -o /tmp/alloc_lottery 2>/dev/null
/tmp/alloc_lottery 8 | grep "round 0" | cut -c1-60
for counters in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum"; do
  name=$(echo "$counters" | tr ' ' '_')
  rm -rf /tmp/lp; rocprofv3 --kernel-trace --pmc $counters -d /tmp/lp -o p -- /tmp/alloc_lottery 8 > /tmp/lp.out 2> /tmp/lp.log
  db=$(find /tmp/lp -name 'p_results.db' | head -1)
  if [ -z "$db" ]; then echo "no db for $counters"; tail -3 /tmp/lp.log; continue; fi
  grep "round 0" /tmp/lp.out | cut -c1-60
  python3 - "$db" <<'PY'
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
per = defaultdict(float)
for k, c, v, d in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
    if "stream128" in k: per[(d, c)] += v
ds = sorted(set(d for d, c in per))
cs = sorted(set(c for d, c in per))
# 30 launches per allocation and round-0 rate call: (2 warm + 8 timed) x 3 rates; print the mean per group of 30 dispatches
for c in cs:
    vals = [per[(d, c)] for d in ds]
    groups = [sum(vals[i:i + 30]) / 30 for i in range(0, min(len(vals), 240), 30)]
    print("  %-34s per allocation: %s" % (c, " ".join("%.4g" % g for g in groups)))
PY
done
