#!/usr/bin/env python3
"""Generates tests/golden/stack_fixture.npz: a small seeded sub-exposure stack
and the oracle's results for every stacking mode (weighted and unweighted).

The reference is Go and cannot run in the build image, so these vectors come
from the C oracle (oracle/nl_oracle.c) after it was checked against
tests/golden/kat.json and against the independent restatement oracle/pyref.py
(this script re-checks the latter before writing).  Run from the repo root:
    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from oracle import oracle, pyref  # noqa: E402
from util import make_frames      # noqa: E402

N, W, H = 24, 20, 9
frames = make_frames(N, W, H, seed=20260929)
weights = (0.2 + 0.8 * np.random.default_rng(42).random(N)).astype(np.float32)
out = {"frames": frames, "weights": weights, "width": np.int32(W), "height": np.int32(H),
       "sigma_low": np.float32(2.5), "sigma_high": np.float32(3.0)}
for mode in range(6):
    for tag, w in (("", None), ("_w", weights)):
        if w is not None and mode in (0, 4, 5):
            continue
        rc, res, cl, ch, _ = oracle.stack_apply(mode, frames, w, 2.5, 3.0)
        assert rc == 0
        r2, cl2, ch2 = pyref.stack(mode, frames, w, 2.5, 3.0)
        assert np.array_equal(res, r2, equal_nan=True) and (cl, ch) == (cl2, ch2), (mode, tag)
        out["mode%d%s" % (mode, tag)] = res
        out["clip%d%s" % (mode, tag)] = np.array([cl, ch], np.int64)
np.savez_compressed(os.path.join(HERE, "stack_fixture.npz"), **out)
print("wrote stack_fixture.npz:", {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("clip") is False})
