"""The hand-derived known answers (tests/golden/kat.json -- K1-K7 and the
order-dependent O1-O4, each traced against internal/qsort/qsort.go:94-126 and
internal/ops/stack/stack.go) and the committed fixture
(tests/golden/stack_fixture.npz) through the C ABI on the GPU.  The oracle is
not consulted here: these pins come from the reference source alone."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
IDS = [c["name"][:44] for c in KAT["cases"]]

# replicate the one-pixel case over a row so that whole waves, the ragged tail of a
# wave and the 4-pixel vector kernels all see it
WIDTH = 131


def _frames(case):
    vals = np.array([np.nan if v is None else v for v in case["values"]], np.float32)
    return np.repeat(vals[:, None], WIDTH, axis=1)


@pytest.mark.parametrize("exact", [0, 1, 2, 3], ids=["dispatch", "exact-lds", "exact-wave", "exact-tile"])
@pytest.mark.parametrize("case", KAT["cases"], ids=IDS)
def test_known_answers_on_the_gpu(nl, case, exact):
    frames = _frames(case)
    mode = case["mode"]
    weighted = "weights" in case
    with nl.StackHandle(frames.shape[0], WIDTH, 1) as st:
        st.upload_frames(frames)
        st.set_weights(np.array(case["weights"], np.float32) if weighted else None)
        st.set_exact(exact)
        got, cl, ch = st.run(mode, *case["sigma"])
        kernel = st.last_kernel_name
    want = np.float32(case["result"])
    assert [cl, ch] == [WIDTH * c for c in case["clip"]], kernel
    # Only the register-resident kernels for UNWEIGHTED sigma / winsor / MAD sum in sorted order
    # (north-star tolerance 1e-5); everything else, and every exact kernel, is bit for bit.
    if exact == 0 and mode in (2, 3, 4) and not weighted:
        assert np.all(np.abs(got.astype(np.float64) - want) <= 1e-5 * abs(float(want))), (kernel, got[:3], want)
    else:
        assert np.all(got == want), (kernel, got[:3], want)
        for what, wrong in case.get("wrong_answers", {}).items():
            assert not np.any(got == np.float32(wrong)), what
    for what, wrong in case.get("wrong_counters", {}).items():
        assert [cl, ch] != [WIDTH * c for c in wrong], what


def test_committed_fixture_on_the_gpu(nl):
    fx = np.load(os.path.join(HERE, "golden", "stack_fixture.npz"))
    frames, weights = fx["frames"], fx["weights"]
    sl, sh = float(fx["sigma_low"]), float(fx["sigma_high"])
    n, p = frames.shape
    checked = 0
    for mode in range(6):
        for tag, w in (("", None), ("_w", weights)):
            if "mode%d%s" % (mode, tag) not in fx:
                continue
            want, clip = fx["mode%d%s" % (mode, tag)], fx["clip%d%s" % (mode, tag)]
            for exact in (1, 2, 3, 0):
                with nl.StackHandle(n, p, 1) as st:
                    st.upload_frames(frames)
                    st.set_weights(w)
                    st.set_exact(exact)
                    got, cl, ch = st.run(mode, sl, sh)
                    kernel = st.last_kernel_name
                assert [cl, ch] == list(clip), (mode, tag, kernel)
                if exact == 0 and mode in (2, 3, 4) and w is None:
                    ok = ~np.isnan(want)
                    assert np.array_equal(np.isnan(got), np.isnan(want))
                    assert np.all(np.abs(got[ok].astype(np.float64) - want[ok]) <= 1e-5 * np.abs(want[ok])), kernel
                else:
                    assert np.array_equal(got, want, equal_nan=True), (mode, tag, kernel)
                checked += 1
    assert checked >= 24
