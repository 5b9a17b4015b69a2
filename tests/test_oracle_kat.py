"""Known answers (tests/golden/kat.json, hand-derived from the reference
source) and the committed fixture (tests/golden/stack_fixture.npz) against BOTH
restatements: the C oracle and the independent numpy-float32 one.  This is the
pin for the Stack* functions, for which the reference has no tests."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def _frames(case):
    vals = [np.nan if v is None else v for v in case["values"]]
    return np.array(vals, np.float32).reshape(-1, 1)


@pytest.mark.parametrize("case", KAT["cases"], ids=[c["name"][:40] for c in KAT["cases"]])
def test_known_answers_c_oracle(oracle, case):
    w = np.array(case["weights"], np.float32) if "weights" in case else None
    rc, res, cl, ch, _ = oracle.stack_apply(case["mode"], _frames(case), w, *case["sigma"])
    assert rc == 0
    assert res[0] == np.float32(case["result"])
    assert [cl, ch] == case["clip"]
    for what, wrong in case.get("wrong_answers", {}).items():          # (the mis-readings on record give other values)
        assert np.float32(wrong) != np.float32(case["result"]), what
    for what, wrong in case.get("wrong_counters", {}).items():
        assert wrong != case["clip"], what


@pytest.mark.parametrize("case", KAT["cases"], ids=[c["name"][:40] for c in KAT["cases"]])
def test_known_answers_python_restatement(case):
    from oracle import pyref
    w = np.array(case["weights"], np.float32) if "weights" in case else None
    res, cl, ch = pyref.stack(case["mode"], _frames(case), w, *case["sigma"])
    assert res[0] == np.float32(case["result"])
    assert [cl, ch] == case["clip"]


def test_linear_regression_quirk(oracle):
    k = KAT["linear_regression"]
    slope, icpt, xm, xsd, ym, ysd = oracle.linear_regression(k["xs"], k["ys"])
    assert slope == np.float32(k["slope"]) and icpt == np.float32(k["intercept"])
    assert (xm, ym) == (np.float32(k["xmean"]), np.float32(k["ymean"]))
    assert (xsd, ysd) == (np.float32(k["xstddev"]), np.float32(k["ystddev"]))


def test_mean_stddev_is_population_std(oracle):
    m, s = oracle.mean_stddev([2, 4, 4, 4, 5, 5, 7, 9])
    assert (m, s) == (np.float32(5), np.float32(2))


def test_committed_fixture_matches_both_restatements(oracle):
    from oracle import pyref
    fx = np.load(os.path.join(HERE, "golden", "stack_fixture.npz"))
    frames, weights = fx["frames"], fx["weights"]
    sl, sh = float(fx["sigma_low"]), float(fx["sigma_high"])
    for mode in range(6):
        for tag, w in (("", None), ("_w", weights)):
            if "mode%d%s" % (mode, tag) not in fx:
                continue
            want, clip = fx["mode%d%s" % (mode, tag)], fx["clip%d%s" % (mode, tag)]
            rc, res, cl, ch, _ = oracle.stack_apply(mode, frames, w, sl, sh)
            assert rc == 0 and np.array_equal(res, want, equal_nan=True) and [cl, ch] == list(clip)
            r2, cl2, ch2 = pyref.stack(mode, frames, w, sl, sh)
            assert np.array_equal(r2, want, equal_nan=True) and [cl2, ch2] == list(clip)
