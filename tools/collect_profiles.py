#!/usr/bin/env python3
"""Copy the summaries tools/gpu_profile.sh left in gpurun_out/ into profiles/ (tracked) under the
round prefix and rebuild profiles/<round>_traffic.json; prints one line per workload.
usage: tools/collect_profiles.py <round, e.g. r02> tag [tag ...]"""
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1]
tags = sys.argv[2:]
entries = []
for t in tags:
    for suffix in ("kernel_stats.txt", "pmc.txt"):
        txt = open("gpurun_out/%s_%s" % (t, suffix)).read().replace("/tmp/code/mlnoga__nightlight/repo/", "")
        open("profiles/%s_%s_%s" % (rnd, t, suffix), "w").write(txt)
    line = [l for l in open("gpurun_out/%s_bench.json" % t) if l.startswith("{")][-1]
    open("profiles/%s_%s_bench.json" % (rnd, t), "w").write(line)
    e = json.loads(open("gpurun_out/%s_traffic.json" % t).read())
    entries.append(e)
    b = json.loads(line)
    r = b["roofline"]
    ks = [l for l in open("gpurun_out/%s_kernel_stats.txt" % t) if r["kernel"] in l]
    avg = float(ks[0].split()[-2]) if ks else -1
    ratio = e["traffic_bytes"] / e["algorithmic_bytes"] if e["algorithmic_bytes"] else 0
    print("%-14s pass %8.3f ms  kernel %-46s %8.3f ms (rocprof avg %8.1f us) frac %.3f pass_frac %.3f  traffic/alg %.4f redo %d"
          % (t, b["ms_per_step"], r["kernel"], r["kernel_ms"], avg, r["frac"], r["pass_frac"], ratio,
             r["pixels_redone_by_exact_kernel"]))
doc = {"_comment": "HBM traffic per launch of the dominant kernel from rocprofv3 PMC passes (profiles/%s_<tag>_pmc.txt)" % rnd + ": "
                   "FETCH_SIZE and WRITE_SIZE in KiB, each collected in its own --pmc run with --kernel-trace only "
                   "(tools/gpu_profile.sh). gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports "
                   "half of the bytes of a coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE "
                   "is taken as is.",
       "entries": entries}
json.dump(doc, open("profiles/%s_traffic.json" % rnd, "w"), indent=1)
if os.path.exists("gpurun_out/bench_default.json"):
    shutil.copy("gpurun_out/bench_default.json", "profiles/%s_bench_default_with_cpu_baseline.json" % rnd)
