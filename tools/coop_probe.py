#!/usr/bin/env python3
"""Developer probe: time the wave-per-pixel exact replay (exact=2) over small
dense tiles, to separate per-pixel work from the scattered-list effects."""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackHandle

for n, w, h in ((128, 4096, 4), (128, 4096, 64), (512, 4096, 4), (512, 4096, 16), (512, 4096, 64)):
    with StackHandle(n, w, h) as st:
        st.fill_synthetic()
        st.set_exact(2)
        for mode in (2, 3):
            st.run(mode, 3.0, 3.0, fetch=False)
            ts = []
            for _ in range(3):
                st.run(mode, 3.0, 3.0, fetch=False)
                ts.append(st.last_kernel_ms)
            print("N=%d %dx%d mode %d  %s  pass ms %s  -> %.1f ns/pixel" %
                  (n, w, h, mode, st.last_kernel_name, ["%.3f" % t for t in ts], 1e6 * min(ts) / (w * h)))
