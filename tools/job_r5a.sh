#!/bin/bash
# round 5, first call: the new extreme-magnitude tests, the whole GPU suite, the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "extreme" > $O/tests_extreme.log 2>&1; echo "rc=$?" >> $O/tests_extreme.log
tail -15 $O/tests_extreme.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; echo "rc=$?" >> $O/tests_all.log
tail -5 $O/tests_all.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
for l in open("gpurun_out/r5a/bench_default.json"):
    if l.startswith("{"):
        b = json.loads(l)
        print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["pass_frac"], b["cpu_baseline"]["parity_with_gpu"])
        for a in b["also"]:
            print(a["tag"], a["ms_per_step"], a["kernel_ms"], a["pass_ms"], a["frac"], a["pass_frac"])
PY
