#!/usr/bin/env python3
"""Developer probe: whole-tile time of the three bit-exact sigma replay engines
(set_exact 1 = LDS lane-per-pixel, 2 = wave-per-pixel, 3 = tile state machine) per frame count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nightlight_amd import StackHandle

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for mode in (2, 3):
    for n in (8, 16, 32, 64, 128, 256):
        w = np.linspace(0.2, 1.0, n).astype(np.float32)
        with StackHandle(n, 4096, rows) as st:
            st.fill_synthetic()
            line = "mode %d N=%3d" % (mode, n)
            for weighted in (False, True):
                st.set_weights(w if weighted else None)
                for ex in (1, 2, 3):
                    st.set_exact(ex)
                    st.run(mode, 3.0, 3.0, fetch=False)
                    st.run(mode, 3.0, 3.0, fetch=False)
                    line += "  %s ex%d %7.2f ms" % ("W" if weighted else "U", ex, st.last_kernel_ms)
            print(line, flush=True)
