#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bench; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; echo "rc=$?" >> $O/tests_all.log
tail -5 $O/tests_all.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
for l in open("gpurun_out/bench/bench_default.json"):
    if l.startswith("{"):
        b = json.loads(l)
        print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["pass_frac"], b["cpu_baseline"]["parity_with_gpu"])
        print(b.get("fresh_handle"))
        for a in b["also"]:
            print(a["tag"], a["ms_per_step"], a["kernel_ms"], a["pass_ms"], a["frac"], a["pass_frac"], (a.get("parity_with_oracle") or {}).get("within_1e-5"), (a.get("parity_with_oracle") or {}).get("clip_counters_equal"))
PY
