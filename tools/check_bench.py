#!/usr/bin/env python3
"""Bench-diff guard: compares a default bench line with profiles/expected_also.json.

    python tools/check_bench.py bench.json            exit 1 if any workload's pass, step or tail is slower than the band allows
    python tools/check_bench.py --update bench.json   rewrite the expectations from this bench line (same commit as the library!)

Per tag: ms_per_step and pass_ms may exceed the expectation by band_pct (8 %: the pool's boxes differ by up to 7 % on the HBM-bound workloads), the tail
pass_ms - kernel_ms by tail_band_pct (10 %) or tail_band_abs_ms, whichever is larger; the C3 tile's goal-seek total by band_pct.
Faster never fails (it is reported so that the expectations get updated).  Rule of the repo since round 6: no library commit
after the last bench that passed this check -- tools/final_check.sh runs build, suite, bench and this script in that order.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPECTED = os.path.join(ROOT, "profiles", "expected_also.json")


def load_bench(path):
    for l in open(path):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("no JSON line in %s" % path)


def entries_of(b):
    out = {"headline": {"ms_per_step": b["ms_per_step"], "pass_ms": b["roofline"]["pass_ms"], "kernel_ms": b["roofline"]["kernel_ms"]}}
    for a in b.get("also", []):
        e = {"ms_per_step": a["ms_per_step"], "pass_ms": a["pass_ms"], "kernel_ms": a["kernel_ms"]}
        if a.get("goal_seek"):
            e["goal_seek_ms"] = a["goal_seek"]["total_ms"]
        out[a["tag"]] = e
    for e in out.values():
        e["tail_ms"] = round(e["pass_ms"] - e["kernel_ms"], 4)
    return out


def main():
    args = sys.argv[1:]
    update = "--update" in args
    args = [a for a in args if a != "--update"]
    if len(args) != 1:
        raise SystemExit(__doc__)
    got = entries_of(load_bench(args[0]))
    exp = json.load(open(EXPECTED))
    if update:
        exp["entries"] = got
        exp["source"] = args[0]
        json.dump(exp, open(EXPECTED, "w"), indent=1)
        print("expectations rewritten from %s (%d tags)" % (args[0], len(got)))
        return 0
    band, tband, tabs = exp["band_pct"] / 100.0, exp["tail_band_pct"] / 100.0, exp["tail_band_abs_ms"]
    bad = 0
    print("%-26s %22s %22s %22s" % ("tag", "ms_per_step got/exp", "pass_ms got/exp", "tail_ms got/exp"))
    for tag, e in exp["entries"].items():
        g = got.get(tag)
        if g is None:
            print("%-26s missing from the bench line" % tag)
            continue
        flags = []
        for k in ("ms_per_step", "pass_ms"):
            if g[k] > e[k] * (1 + band):
                flags.append("%s +%.1f %%" % (k, (g[k] / e[k] - 1) * 100))
            elif g[k] < e[k] * (1 - band):
                flags.append("(%s %.1f %%: faster, update the expectation)" % (k, (g[k] / e[k] - 1) * 100))
        if g["tail_ms"] > max(e["tail_ms"] * (1 + tband), e["tail_ms"] + tabs):
            flags.append("tail +%.3f ms" % (g["tail_ms"] - e["tail_ms"]))
        if "goal_seek_ms" in e and g.get("goal_seek_ms", 0) > e["goal_seek_ms"] * (1 + band):
            flags.append("goal-seek %.1f vs %.1f ms" % (g["goal_seek_ms"], e["goal_seek_ms"]))
        slow = [f for f in flags if not f.startswith("(")]
        bad += bool(slow)
        print("%-26s %10.3f / %-9.3f %10.3f / %-9.3f %10.3f / %-9.3f %s" % (
            tag, g["ms_per_step"], e["ms_per_step"], g["pass_ms"], e["pass_ms"], g["tail_ms"], e["tail_ms"],
            ("  <-- " if slow else "  ") + "; ".join(flags)))
    print("REGRESSION in %d workload(s)" % bad if bad else "ok: no workload slower than its band")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
