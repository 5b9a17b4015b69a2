#!/bin/bash
# full GPU validation: every -m gpu test, the same with chunked passes forced, smoke, default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/full; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; echo "rc=$?" >> $O/tests_all.log
tail -3 $O/tests_all.log
NL_CHUNKS=40,30,20,10 timeout 900 python -m pytest tests -m gpu -x -q -k "not fullsize and not dist" > $O/tests_forced.log 2>&1; echo "forced rc=$?" >> $O/tests_forced.log
tail -2 $O/tests_forced.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python3 - <<'PY'
import json
for l in open("gpurun_out/full/bench_default.json"):
    if l.startswith("{"):
        b = json.loads(l)
        print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["pass_frac"], b["cpu_baseline"]["parity_with_gpu"]["clip_counters_equal"], b["cpu_baseline"]["parity_with_gpu"]["within_1e-5"])
        print(b.get("fresh_handle"))
        for a in b["also"]:
            print(a["tag"], a["ms_per_step"], a["kernel_ms"], a["pass_ms"], a["frac"], a["pass_frac"], (a.get("parity_with_oracle") or {}).get("within_1e-5"), (a.get("parity_with_oracle") or {}).get("clip_counters_equal"))
PY
