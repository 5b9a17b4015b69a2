"""Host time of nl_group_run with the result downloaded (run on the GPU box): NL_GROUP_PARALLEL_FINISH=0 / 1.
    python tools/group_run_probe.py [tiles] [frames]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackGroup

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for flag in ("0", "1", "0", "1"):
    os.environ["NL_GROUP_PARALLEL_FINISH"] = flag
    with StackGroup(n, 4096, 4096, devices=[0] * tiles) as g:
        g.fill_synthetic(1)
        for _ in range(3):
            g.run(2, 3.0, 3.0)
        t = []
        for _ in range(10):
            t0 = time.perf_counter()
            g.run(2, 3.0, 3.0)
            t.append((time.perf_counter() - t0) * 1e3)
        print("group of %d tiles, %d x 4096^2, finish on worker threads %s: run + download %.3f ms (median of 10, min %.3f)"
              % (tiles, n, flag, float(np.median(t)), min(t)))
