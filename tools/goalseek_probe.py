#!/usr/bin/env python3
"""Configuration C3 on one of its 8 tiles: 512 x (512 x 4096) frames, winsorized sigma clip,
bisection on the clip percentages (stackfindsigma.go:48-98) -- passes and wall time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightlight_amd.stack import StackHandle

for n, rows, mode in ((512, 512, 3), (128, 4096, 2)):
    with StackHandle(n, 4096, rows) as st:
        st.fill_synthetic()
        st.find_sigmas(mode, 0.5, 0.5)
        t0 = time.perf_counter()
        out, cl, ch, sl, sh, passes = st.find_sigmas(mode, 0.5, 0.5)
        dt = time.perf_counter() - t0
        print("mode %d, %d x %dx4096: goal-seek 0.5 %% / 0.5 %% -> sigma %.4f / %.4f after %d passes, %.1f ms (%.2f ms per pass), "
              "clipped %d low %d high" % (mode, n, rows, sl, sh, passes, dt * 1e3, dt * 1e3 / passes, cl, ch))
