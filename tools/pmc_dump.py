#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from a rocprofv3 results .db (developer utility)."""
import sqlite3
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if not view:
        print(path, "no counters_collection view; tables:", [t for t in tabs if "pmc" in t or "counter" in t])
        continue
    cols = [d[1] for d in cur.execute("pragma table_info(%s)" % view)]
    acc = defaultdict(lambda: defaultdict(list))
    q = "select kernel_name, counter_name, value, dispatch_id from %s" % view
    per = defaultdict(float)
    for k, c, v, d in cur.execute(q):
        per[(k, c, d)] += v
    for (k, c, d), v in per.items():
        acc[k][c].append(v)
    print("==", path)
    for k, cs in acc.items():
        print(k[:70])
        for c, vs in sorted(cs.items()):
            print("    %-24s n=%d avg=%.4g" % (c, len(vs), sum(vs) / len(vs)))
