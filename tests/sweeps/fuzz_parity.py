#!/usr/bin/env python3
"""(Lives under tests/ because it uses the CPU oracle.)
Randomised differential test (run on the GPU box): random frame counts, modes, sigmas, NaN
fractions, ties, infinities, tile geometry and frame stride through the default dispatch of the C ABI against
the oracle.  Counters must be identical, values bit-exact or within 1e-5 depending on the kernel.
usage: fuzz_parity.py [cases] [seed]
NL_FUZZ_N=lo,hi restricts the frame counts, NL_FUZZ_MODES=2,3 the modes (a kernel class under test),
NL_FUZZ_WEIGHTED=p sets the share of weighted cases (default 0.25)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_frames
from nightlight_amd.stack import StackHandle
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
protocols = {}            # second passes by nl_stack_last_pass_protocol
t0 = time.time()
for i in range(cases):
    mode = int(rng.choice([0, 1, 2, 2, 3, 3, 4, 5, 5]))
    n = int(rng.choice([rng.integers(1, 33), rng.integers(33, 129), rng.integers(129, 513), rng.integers(513, 700)],
                       p=[0.35, 0.35, 0.25, 0.05]))
    if os.environ.get("NL_FUZZ_N"):
        lo_n, hi_n = (int(x) for x in os.environ["NL_FUZZ_N"].split(","))
        n = int(rng.integers(lo_n, hi_n + 1))
    if os.environ.get("NL_FUZZ_MODES"):
        mode = int(rng.choice([int(x) for x in os.environ["NL_FUZZ_MODES"].split(",")]))
    width, height = int(rng.integers(3, 150)), int(rng.integers(1, 12))
    row0 = int(rng.integers(0, height))
    rows = int(rng.integers(1, height - row0 + 1))
    sl, sh = float(np.float32(rng.uniform(0.3, 4.5))), float(np.float32(rng.uniform(0.3, 4.5)))
    nan_frac = float(rng.choice([0.0, 0.002, 0.02, 0.2]))
    frames = make_frames(n, width, height, seed=int(rng.integers(1 << 30)), nan_frac=nan_frac,
                         ties=bool(rng.integers(2)), nan_border=bool(rng.integers(2)))
    if rng.random() < 0.3 and mode != 4:
        frames[int(rng.integers(n)), int(rng.integers(width * height))] = np.inf * (1 if rng.random() < 0.5 else -1)
    r = rng.random()
    if r < 0.06:
        sl = float(rng.choice([-1.0, 0.0, 0.01, 20.0]))       # degenerate bounds (negative sigma: quirk Q6)
    elif r < 0.12:
        sh = float(rng.choice([-1.0, 0.0, 0.01, 20.0]))
    r = rng.random()
    if r < 0.05:
        frames = (frames * np.float32(1e-36)).astype(np.float32)      # subnormal variances
    elif r < 0.10:
        frames = (frames * np.float32(1e30)).astype(np.float32)       # squares overflow
    elif r < 0.15:
        frames[:, : width * height // 3] = np.float32(rng.uniform(-5, 5))   # constant pixels
    weights = None
    if mode in (1, 2, 3) and rng.random() < float(os.environ.get("NL_FUZZ_WEIGHTED", "0.25")):
        weights = rng.uniform(0.2, 1.0, n).astype(np.float32)
    ref_loc = float(rng.choice([0.0, 7.5]))
    # frame stride of the owned buffer (read per handle, nlstack_api.hip padded_frame_stride): dense or one of a few paddings
    os.environ["NL_STRIDE_PAD"] = str(int(rng.choice([0, 0, 4, 64, 1092, 16448])))
    with StackHandle(n, width, height, row0=row0, rows=rows) as st:
        st.upload_frames(frames)
        st.set_weights(weights)
        got, cl, ch = st.run(mode, sl, sh, ref_loc)
        kernel = st.last_kernel_name
        again_ok = True
        if mode in (2, 3):
            # a second pass on the handle knows its list lengths: fused protocol, generic pass + first replay in one launch
            # where that applies -- it must end exactly like the first
            got2, cl2, ch2 = st.run(mode, sl, sh, ref_loc)
            again_ok = (cl2, ch2) == (cl, ch) and np.array_equal(got2.view(np.uint32), got.view(np.uint32))
            kernel += " protocol %d" % st.last_pass_protocol
            protocols[st.last_pass_protocol] = protocols.get(st.last_pass_protocol, 0) + 1
    got = got[row0 * width:(row0 + rows) * width]
    tile = np.ascontiguousarray(frames.reshape(n, height, width)[:, row0:row0 + rows, :].reshape(n, -1))
    ow = None if mode in (0, 5) else weights
    rc, want, wl, wh, _ = oracle.stack_apply(mode, tile, ow, sl, sh, ref_loc, num_cpu=4)
    same_nan = np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want) & (want != got)
    with np.errstate(all="ignore"):
        rel = float(np.nanmax(np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok].astype(np.float64)))) if ok.any() else 0.0
    good = rc == 0 and same_nan and (rel <= 1e-5) and (mode < 2 or (cl, ch) == (wl, wh)) and again_ok
    if not good:
        bad += 1
        print("FAIL case %d: mode %d n=%d %dx%d rows[%d,%d) sl=%r sh=%r nan=%g weights=%s pad=%s kernel=%s counters %r vs %r rel %.3g same_nan %s second_pass_same %s"
              % (i, mode, n, width, height, row0, row0 + rows, sl, sh, nan_frac, weights is not None, os.environ["NL_STRIDE_PAD"], kernel,
                 (cl, ch), (wl, wh), rel, same_nan, again_ok), flush=True)
print("fuzz: %d cases, %d failing, %.0f s; second passes by protocol (bit 0 fused, bit 1 one-launch tail): %r" % (cases, bad, time.time() - t0, protocols))
sys.exit(1 if bad else 0)
