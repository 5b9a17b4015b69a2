"""A/B timing of frame-stride paddings inside ONE process (run on the GPU box):
    python tools/ab_stride.py <pads> <workload> [<workload> ...]
pads: comma list of NL_STRIDE_PAD values (floats added to the tile's pixel count); "d" = the library's default rule.
workload: mode:frames:rows[:image_rows[:w]] (w = weighted), 4096 pixels wide.
One handle per padding (the variable is read at create), passes interleaved (reps rounds of 3 warm-up + 10 timed per
handle); prints the median over the rounds of the mean pass / dominant-kernel time."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd import StackHandle

pads = sys.argv[1].split(",")
reps = int(os.environ.get("AB_REPS", "3"))
for wl in sys.argv[2:]:
    f = wl.split(":")
    mode, n, rows = int(f[0]), int(f[1]), int(f[2])
    image_rows = int(f[3]) if len(f) > 3 and f[3] else rows
    weighted = len(f) > 4 and f[4] == "w"
    hs = []
    for p in pads:
        if p == "d":
            os.environ.pop("NL_STRIDE_PAD", None)
        else:
            os.environ["NL_STRIDE_PAD"] = p
        st = StackHandle(n, 4096, image_rows, device=0, row0=0, rows=rows)
        st.fill_synthetic(seed=1)
        if weighted:
            st.set_weights(np.random.default_rng(5).uniform(0.5, 1.5, n).astype(np.float32))
        hs.append(st)
    res = [[] for _ in pads]
    for rep in range(reps):
        for i, st in enumerate(hs):
            for _ in range(3):
                st.run_async(mode, 3.0, 3.0, 0.0)
            st.finish()
            for _ in range(10):
                st.run_async(mode, 3.0, 3.0, 0.0)
            cl, ch = st.finish()
            t = [st.pass_times(b) for b in range(10)]
            res[i].append((float(np.mean([x[0] for x in t])), float(np.mean([x[1] for x in t])), cl, ch))
    for i, p in enumerate(pads):
        r = res[i]
        print("%-18s pad %-8s stride %10d: pass %8.4f ms (min %8.4f)  dominant %8.4f ms  clips %d/%d  %s"
              % (wl, p, hs[i].frame_stride(), np.median([x[0] for x in r]), min(x[0] for x in r),
                 np.median([x[1] for x in r]), r[-1][2], r[-1][3], hs[i].last_kernel_name), flush=True)
    for st in hs:
        st.close()
