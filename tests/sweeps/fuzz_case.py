#!/usr/bin/env python3
"""Re-create one case of fuzz_parity.py (same seed, same draw order) and bisect it to the pixels whose
clip counters differ from the oracle: every pixel is stacked on its own (1x1 tile) through the default
dispatch.  usage: fuzz_case.py <seed> <case index>   (run on the GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_frames
from nightlight_amd.stack import StackHandle
from oracle import oracle

seed, want_case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for i in range(want_case + 1):
    mode = int(rng.choice([0, 1, 2, 2, 3, 3, 4, 5, 5]))
    n = int(rng.choice([rng.integers(1, 33), rng.integers(33, 129), rng.integers(129, 513), rng.integers(513, 700)],
                       p=[0.35, 0.35, 0.25, 0.05]))
    width, height = int(rng.integers(3, 150)), int(rng.integers(1, 12))
    row0 = int(rng.integers(0, height))
    rows = int(rng.integers(1, height - row0 + 1))
    sl, sh = float(np.float32(rng.uniform(0.3, 4.5))), float(np.float32(rng.uniform(0.3, 4.5)))
    nan_frac = float(rng.choice([0.0, 0.002, 0.02, 0.2]))
    mf_seed = int(rng.integers(1 << 30)); ties = bool(rng.integers(2)); nb = bool(rng.integers(2))
    frames = make_frames(n, width, height, seed=mf_seed, nan_frac=nan_frac, ties=ties, nan_border=nb) if i == want_case else None
    if rng.random() < 0.3 and mode != 4:
        a, b, c = int(rng.integers(n)), int(rng.integers(width * height)), rng.random()
        if frames is not None:
            frames[a, b] = np.inf * (1 if c < 0.5 else -1)
    r = rng.random()
    if r < 0.06:
        sl = float(rng.choice([-1.0, 0.0, 0.01, 20.0]))
    elif r < 0.12:
        sh = float(rng.choice([-1.0, 0.0, 0.01, 20.0]))
    r = rng.random()
    if r < 0.05:
        if frames is not None: frames = (frames * np.float32(1e-36)).astype(np.float32)
    elif r < 0.10:
        if frames is not None: frames = (frames * np.float32(1e30)).astype(np.float32)
    elif r < 0.15:
        v = np.float32(rng.uniform(-5, 5))
        if frames is not None: frames[:, : width * height // 3] = v
    weights = None
    if mode in (1, 2, 3) and rng.random() < 0.25:
        weights = rng.uniform(0.2, 1.0, n).astype(np.float32)
    ref_loc = float(rng.choice([0.0, 7.5]))
print("case %d: mode %d n=%d %dx%d rows[%d,%d) sl=%r sh=%r nan=%g ties=%s weights=%s" % (want_case, mode, n, width, height, row0, row0 + rows, sl, sh, nan_frac, ties, weights is not None))
tile = np.ascontiguousarray(frames.reshape(n, height, width)[:, row0:row0 + rows, :].reshape(n, -1))
ow = None if mode in (0, 5) else weights
bad = 0
for p in range(tile.shape[1]):
    col = np.ascontiguousarray(tile[:, p:p + 1])
    with StackHandle(n, 1, 1) as st:
        st.upload_frames(col)
        st.set_weights(weights)
        got, cl, ch = st.run(mode, sl, sh, ref_loc)
        kernel, gen, fb = st.last_kernel_name, st.last_generic_pixels, st.last_fallback_pixels
    rc, want, wl, wh, _ = oracle.stack_apply(mode, col, ow, sl, sh, ref_loc, num_cpu=1)
    if (cl, ch) != (wl, wh) or not np.array_equal(got, want, equal_nan=True):
        bad += 1
        x = col[:, 0]; x = np.sort(x[~np.isnan(x)])
        print("pixel %d: counters %r vs oracle %r, value %r vs %r, %d valid samples, generic %d exact %d (%s)" % (p, (cl, ch), (wl, wh), got[0], want[0], x.size, gen, fb, kernel))
        print("  sorted samples:", np.array2string(x, max_line_width=200, precision=9, floatmode="unique"))
print("%d pixel(s) differ" % bad)
