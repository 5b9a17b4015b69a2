"""Host time of nl_group_create / nl_group_destroy (run on the GPU box): the drop-in makes and drops a group per Apply.
    python tools/group_create_probe.py [tiles] [frames]      (all tiles on device 0 when the box has one GPU)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackGroup

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tc, td = [], []
for rep in range(8):
    t0 = time.perf_counter()
    g = StackGroup(n, 4096, 4096, devices=[0] * tiles)
    t1 = time.perf_counter()
    g.close()
    t2 = time.perf_counter()
    tc.append((t1 - t0) * 1e3)
    td.append((t2 - t1) * 1e3)
print("group of %d tiles, %d x 4096^2: create %.3f ms first / %.3f ms median of the rest; destroy %.3f / %.3f"
      % (tiles, n, tc[0], float(np.median(tc[1:])), td[0], float(np.median(td[1:]))))
