#!/usr/bin/env python3
"""(Lives under tests/ because it uses the CPU oracle, which only tests, smoke() and the
cpu_baseline leg of bench.py may touch.)
Large-sample parity sweep (run on the GPU box): several synthetic seeds and frame counts,
the default dispatch through the C ABI against the CPU oracle on every pixel.  Clip counters
must be identical; values bit-exact for the exact kernels, within 1e-5 for the register
kernels.  Rare-event coverage (decisions with probabilities around 1e-7 per sample)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nightlight_amd.stack import StackHandle
from oracle import oracle

CASES = [  # mode, frames, width, rows, weighted
    (2, 128, 4096, 1024, False), (3, 128, 4096, 512, False), (2, 512, 4096, 256, False), (3, 512, 4096, 128, False),
    (2, 300, 4096, 256, False), (3, 200, 4096, 256, False), (2, 32, 4096, 2048, False), (3, 24, 4096, 1024, False),
    (2, 64, 4096, 1024, False), (5, 128, 4096, 256, False), (0, 64, 4096, 1024, False), (2, 100, 4096, 512, False),
    (2, 128, 4096, 128, True), (3, 96, 4096, 128, True),
    (4, 100, 4096, 512, False), (4, 256, 4096, 128, False), (5, 256, 4096, 64, False), (5, 400, 4096, 32, False),
    (0, 512, 4096, 128, False), (0, 200, 4096, 256, False), (2, 600, 4096, 32, False), (0, 600, 4096, 32, False),
    (4, 128, 4096, 512, False), (4, 120, 4096, 256, False), (5, 120, 4096, 128, False), (3, 100, 4096, 256, False),
]
bad = 0
for seed in (11, 12, 13):
    for mode, n, w, rows, weighted in CASES:
        t0 = time.time()
        with StackHandle(n, w, rows) as st:
            st.fill_synthetic(seed)
            frames = np.stack([st.download_tile(i) for i in range(n)])
            weights = None
            if weighted:
                weights = np.random.default_rng(seed).uniform(0.2, 1.0, n).astype(np.float32)
                st.set_weights(weights)
            got, cl, ch = st.run(mode, 3.0, 2.5)
            kernel = st.last_kernel_name
            redo = st.last_fallback_pixels
        ow = None if mode in (0, 5) else weights
        rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, ow, 3.0, 2.5, 0.0, num_cpu=os.cpu_count())
        same_nan = np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want) & (want != got)
        rel = float(np.max(np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok].astype(np.float64)))) if ok.any() else 0.0
        good = rc == 0 and same_nan and rel <= 1e-5 and (mode < 2 or (cl, ch) == (wl, wh))
        bad += 0 if good else 1
        print("%s seed %d mode %d n=%3d %dx%d%s  %-46s counters %s max_rel %.2e replayed %d  (%.1f s)"
              % ("ok  " if good else "FAIL", seed, mode, n, w, rows, " w" if weighted else "", kernel,
                 "equal" if (cl, ch) == (wl, wh) else "%r vs %r" % ((cl, ch), (wl, wh)), rel, redo, time.time() - t0), flush=True)
print("parity sweep: %d failing case(s)" % bad)
sys.exit(1 if bad else 0)
