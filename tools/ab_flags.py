"""A/B timing of the fast path's developer switches inside ONE process (run on the GPU box):
    python tools/ab_flags.py <mode> <frames> <rows> [row0] [image_rows] [reps] [flags,flags,...]
Variants are interleaved (reps rounds of: each variant, 3 warm-up + 10 timed passes queued back to back), so
clock drift of the box hits all of them alike; prints the median over the rounds of the mean pass / dominant
kernel time per variant.  flags: nl_stack_set_dev_flags (1 = plain pass protocol, 2 = replay in front of the generic pass)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
row0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
image_rows = int(sys.argv[5]) if len(sys.argv) > 5 else rows
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
variants = [int(x) for x in sys.argv[7].split(",")] if len(sys.argv) > 7 else [0, 1, 2, 3]
with StackHandle(n, 4096, image_rows, device=0, row0=row0, rows=rows) as st:
    st.fill_synthetic(seed=1)
    res = {v: [] for v in variants}
    for rep in range(reps):
        for v in variants:
            st.set_dev_flags(v)
            for _ in range(3):
                st.run_async(mode, 3.0, 3.0, 0.0)
            st.finish()
            for _ in range(10):
                st.run_async(mode, 3.0, 3.0, 0.0)
            cl, ch = st.finish()
            t = [st.pass_times(b) for b in range(10)]
            res[v].append((float(np.mean([x[0] for x in t])), float(np.mean([x[1] for x in t])), cl, ch,
                           st.last_generic_pixels, st.last_fallback_pixels))
    for v in variants:
        r = res[v]
        print("mode %d frames %d rows %d flags %d: pass %.4f ms (min %.4f)  dominant %.4f ms (min %.4f)  clips %d/%d generic %d exact %d  %s"
              % (mode, n, rows, v, np.median([x[0] for x in r]), min(x[0] for x in r), np.median([x[1] for x in r]),
                 min(x[1] for x in r), r[-1][2], r[-1][3], r[-1][4], r[-1][5], st.last_kernel_name))
