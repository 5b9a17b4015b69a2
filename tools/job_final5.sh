#!/bin/bash
# round 5, final library: soak (repeatability), every-pixel parity sweep, whole GPU suite, default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final5; mkdir -p $O
timeout 1200 python tools/soak.py > $O/soak.log 2>&1; tail -12 $O/soak.log
python - <<'PY' > gpurun_out/final5/soak_winsor.log 2>&1
import sys, zlib
sys.path.insert(0, ".")
import numpy as np
from nightlight_amd.stack import StackHandle
for n, rows, passes in ((16, 1024, 60), (24, 1024, 60), (40, 512, 40), (96, 512, 30)):
    with StackHandle(n, 4096, rows) as st:
        st.fill_synthetic(5)
        seen = set()
        for _ in range(passes):
            out, cl, ch = st.run(3, 3.0, 3.0)
            seen.add((cl, ch, zlib.crc32(out.tobytes())))
        print("%s winsorized n=%d rows=%d: %d passes, %d distinct results" % ("ok  " if len(seen) == 1 else "FAIL", n, rows, passes, len(seen)))
PY
cat $O/soak_winsor.log | grep -v amdgpu
timeout 2400 python tests/sweeps/parity_sweep.py > $O/parity_sweep.log 2>&1; tail -6 $O/parity_sweep.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; echo "rc=$?" >> $O/tests_all.log; tail -3 $O/tests_all.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
for l in open("gpurun_out/final5/bench_default.json"):
    if l.startswith("{"):
        b = json.loads(l)
        print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["pass_frac"], b["roofline"]["traffic"], b["cpu_baseline"]["value"], b["cpu_baseline"]["parity_with_gpu"]["clip_counters_equal"], b["cpu_baseline"]["parity_with_gpu"]["within_1e-5"])
        for a in b["also"]:
            print(a["tag"], a["ms_per_step"], a["kernel_ms"], a["frac"], a["pass_frac"], a.get("traffic"), (a.get("parity_with_oracle") or {}).get("within_1e-5"), (a.get("parity_with_oracle") or {}).get("clip_counters_equal"))
PY
