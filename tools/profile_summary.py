#!/usr/bin/env python3
"""Turn rocprofv3 result databases into the small text summaries committed
under profiles/ (kernel-trace stats and per-kernel PMC averages)."""
import sqlite3
import sys
from collections import defaultdict


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    out = ["# rocprofv3 --kernel-trace --stats  (durations in us)  source: %s" % path,
           "%-88s %6s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%")]
    for name, calls, total, avg, pct in rows:
        out.append("%-88s %6d %12.1f %10.1f %6.2f" % (name[:88], calls, total, avg, pct))
    return "\n".join(out)


def pmc(path):
    db = sqlite3.connect(path)
    per = defaultdict(float)
    for k, c, v, d in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        per[(k, c, d)] += v
    acc = defaultdict(lambda: defaultdict(list))
    for (k, c, d), v in per.items():
        acc[k][c].append(v)
    out = ["# rocprofv3 --pmc  (per-dispatch sums over all XCDs/SEs, averaged over dispatches)  source: %s" % path]
    for k, cs in acc.items():
        if not k.startswith(("void nl::", "nl::")):
            continue
        out.append(k[:100])
        for c, vs in sorted(cs.items()):
            out.append("    %-24s dispatches=%d avg=%.6g" % (c, len(vs), sum(vs) / len(vs)))
    return "\n".join(out)


def traffic(pmc_txt, bench_json):
    """One entry for profiles/r01_traffic.json: HBM bytes per launch of the
    bench line's dominant kernel.  gfx950: FETCH_SIZE (KiB) counts half of a
    coalesced streaming read (MI355X_MICROARCH.md, HBM section)."""
    import json
    line = [l for l in open(bench_json) if l.startswith("{")][-1]
    bench = json.loads(line)
    roof, cfg = bench["roofline"], bench["config"]
    base = roof["kernel"]                       # as rocprofv3 prints it, minus namespace and arguments
    best = {}
    kernel = None
    for l in open(pmc_txt):
        if not l.startswith(" "):
            kernel = l.strip()
            continue
        f = l.split()
        if kernel and base in kernel and f[0] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
            v = float(f[2].split("=")[1])
            if v > best.get((f[0]), (0.0, ""))[0]:
                best[f[0]] = (v, kernel)
    fetch, write = best.get("FETCH_SIZE", (0.0, ""))[0], best.get("WRITE_SIZE", (0.0, ""))[0]
    entry = {"kernel": roof["kernel"], "frames": cfg["frames"], "width": cfg["width"],
             "rows": cfg["rows_per_gpu"], "mode": cfg["mode"],
             "fetch_size_kib": fetch, "write_size_kib": write,
             "traffic_bytes": 2.0 * fetch * 1024.0 + write * 1024.0,
             "algorithmic_bytes": roof["algorithmic_bytes"],
             # wave-instructions per launch (issue-rate roofline of bench.py, round 6)
             "insts_valu": best.get("SQ_INSTS_VALU", (0.0, ""))[0], "insts_salu": best.get("SQ_INSTS_SALU", (0.0, ""))[0],
             "profiled_kernel": best.get("FETCH_SIZE", (0.0, ""))[1]}
    return json.dumps(entry)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "traffic":
        print(traffic(path, sys.argv[3]))
    else:
        print(kernel_stats(path) if mode == "stats" else pmc(path))
