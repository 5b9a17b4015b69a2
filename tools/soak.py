#!/usr/bin/env python3
"""Repeatability soak (run on the GPU box): many passes of the same stack must give the same
counters and the same output bits every time (hand-over lists, side-stream overlap, cascades
and atomics are the places where a race would show)."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackHandle

bad = 0
for mode, n, rows, passes, weighted in ((2, 128, 1024, 150, False), (3, 128, 512, 60, False), (5, 128, 256, 40, False),
                                       (2, 512, 128, 60, False), (3, 300, 128, 30, False), (0, 64, 1024, 60, False),
                                       (4, 96, 512, 40, False), (5, 256, 64, 20, False), (2, 64, 256, 20, True)):
    with StackHandle(n, 4096, rows) as st:
        st.fill_synthetic(5)
        if weighted:
            st.set_weights(np.linspace(0.2, 1.0, n).astype(np.float32))
        seen = set()
        for _ in range(passes):
            out, cl, ch = st.run(mode, 3.0, 3.0)
            seen.add((cl, ch, zlib.crc32(out.tobytes())))
        ok = len(seen) == 1
        bad += 0 if ok else 1
        print("%s mode %d n=%d rows=%d%s: %d passes, %d distinct results  %s" %
              ("ok  " if ok else "FAIL", mode, n, rows, " weighted" if weighted else "", passes, len(seen), st.last_kernel_name),
              flush=True)
print("soak: %d failing case(s)" % bad)
sys.exit(1 if bad else 0)
