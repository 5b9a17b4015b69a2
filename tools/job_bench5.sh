#!/bin/bash
# round 5: the dist tests (bench's multi-rank path) and the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/b5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > $O/tests_dist.log 2>&1; echo "rc=$?" >> $O/tests_dist.log
tail -3 $O/tests_dist.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
tail -3 $O/bench_default.err
python3 - <<'PY'
import json
for l in open("gpurun_out/b5/bench_default.json"):
    if l.startswith("{"):
        b = json.loads(l)
        print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["pass_frac"], b["cpu_baseline"]["parity_with_gpu"])
        print(json.dumps(b.get("fresh_handle"))[:1500])
        for a in b["also"]:
            print(a["tag"], a["ms_per_step"], a["kernel_ms"], a["pass_ms"], a["frac"], a["pass_frac"], a.get("traffic"), (a.get("parity_with_oracle") or {}).get("within_1e-5"), (a.get("parity_with_oracle") or {}).get("clip_counters_equal"))
PY
timeout 600 python bench.py --gpus 2 --share-device --backend gloo --steps 5 --warmup 2 --preheat-steps 4 --no-also --no-cpu --frames 32 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('2 ranks:', d['value'], d['n_gpus'], d.get('ranks'))
"
