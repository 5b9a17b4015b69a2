"""The two independent restatements (C: oracle/nl_oracle.c, Python:
oracle/pyref.py) must agree bit for bit on seeded random stacks with NaNs,
outliers and ties, for every mode; plus properties of OpStack.Apply
(internal/ops/stack/stack.go:115-227)."""
import numpy as np
import pytest

from util import make_frames


@pytest.mark.parametrize("mode", range(6))
def test_c_and_python_restatements_agree(oracle, mode):
    from oracle import pyref
    rng = np.random.default_rng(100 + mode)
    for trial in range(12):
        n = int(rng.integers(1, 48))
        frames = make_frames(n, 7, 3, seed=1000 * mode + trial, ties=(trial % 3 == 0))
        weights = (0.2 + rng.random(n)).astype(np.float32)
        for w in (None, weights):
            if w is not None and mode in (0, 4, 5):
                continue
            s = float(rng.choice([1.0, 2.0, 2.75, 3.0]))
            rc, res, cl, ch, _ = oracle.stack_apply(mode, frames, w, s, s + 0.5)
            r2, cl2, ch2 = pyref.stack(mode, frames, w, s, s + 0.5)
            assert rc == 0
            assert np.array_equal(res, r2, equal_nan=True), (mode, trial)
            assert (cl, ch) == (cl2, ch2)


def test_apply_result_does_not_depend_on_threads_or_batching(oracle):
    frames = make_frames(20, 64, 33, seed=5)
    base = oracle.stack_apply(3, frames, None, 2.5, 2.5, num_cpu=1)
    for cpus in (2, 8):
        got = oracle.stack_apply(3, frames, None, 2.5, 2.5, num_cpu=cpus)
        assert np.array_equal(base[1], got[1], equal_nan=True) and base[2:4] == got[2:4]


def test_auto_mode_thresholds(oracle):
    # stack.go:45-55
    assert [oracle.auto_select_mode(n) for n in (1, 5, 6, 14, 15, 24, 25, 500)] == [1, 1, 2, 2, 3, 3, 5, 5]


def test_error_codes(oracle):
    frames = make_frames(4, 4, 2, seed=1)
    assert oracle.stack_apply(7, frames)[0] == oracle.ERR_INVALID_MODE
    assert oracle.stack_apply(-1, frames)[0] == oracle.ERR_INVALID_MODE
    assert oracle.stack_apply(4, frames, np.ones(4, np.float32))[0] == oracle.ERR_WEIGHTED_MAD


def test_nan_frame_equals_dropped_frame(oracle):
    frames = make_frames(12, 16, 4, seed=9, nan_frac=0.0, nan_border=False, all_nan_patch=False)
    with_nan = frames.copy()
    with_nan[5] = np.nan
    dropped = np.delete(frames, 5, axis=0)
    for mode in range(6):
        a = oracle.stack_apply(mode, with_nan, None, 2.0, 2.0)
        b = oracle.stack_apply(mode, dropped, None, 2.0, 2.0)
        assert np.array_equal(a[1], b[1]) and a[2:4] == b[2:4]


def test_huge_sigma_is_plain_mean(oracle):
    frames = make_frames(9, 16, 4, seed=2)
    mean = oracle.stack_apply(1, frames)[1]
    for mode in (2, 3):
        res = oracle.stack_apply(mode, frames, None, 1e9, 1e9)
        # same samples, but summed in quickselect order instead of frame order
        assert np.allclose(res[1], mean, rtol=2e-6, atol=0, equal_nan=True) and res[2:4] == (0, 0)


def test_get_weights(oracle):
    # stack.go:231-270
    rc, w, bad = oracle.get_weights(oracle.WEIGHT_NONE, [1, 2, 3])
    assert rc == 0 and w is None
    rc, w, bad = oracle.get_weights(oracle.WEIGHT_EXPOSURE, [30, 60, 0])
    assert rc == oracle.ERR_MISSING_EXPOSURE and bad == 2
    rc, w, bad = oracle.get_weights(oracle.WEIGHT_EXPOSURE, [30, 60])
    assert rc == 0 and list(w) == [30, 60]
    rc, w, bad = oracle.get_weights(oracle.WEIGHT_INVERSE_NOISE, [2.0, 3.0, 4.0])
    assert rc == 0 and np.allclose(w, [1.0, 1 / 3.0, 0.2])
    rc, w, bad = oracle.get_weights(oracle.WEIGHT_INVERSE_HFR, [5.0, 5.0])     # 0/0 -> NaN, as the reference
    assert rc == 0 and np.isnan(w).all()
    assert oracle.get_weights(9, [1.0])[0] == oracle.ERR_INVALID_WEIGHTING


def test_stack_incremental(oracle):
    # stack.go:924-944: frame-count weighted stack of stacks
    rng = np.random.default_rng(4)
    a, b = rng.random(50).astype(np.float32), rng.random(50).astype(np.float32)
    acc = oracle.stack_incremental(np.zeros(50, np.float32), a, 3.0, first=True)
    acc = oracle.stack_incremental(acc, b, 5.0, first=False)
    acc = oracle.stack_incremental_finalize(acc, 8.0)
    want = (a * np.float32(3) + b * np.float32(5)) * (np.float32(1) / np.float32(8))
    assert np.array_equal(acc, want.astype(np.float32))


def test_goal_seek_bisection(oracle):
    # stackfindsigma.go:48-98: clip percentages converge onto the 0.01 % grid or stop after 21 passes
    frames = make_frames(24, 64, 32, seed=77, nan_frac=0.0, nan_border=False, all_nan_patch=False)
    passes, res, cl, ch, sl, sh = oracle.find_sigmas_bisect(2, frames, 0.5, 0.5, num_cpu=4)
    total = frames.size
    assert 1 <= passes <= 21 and 1.0 <= sl <= 11.0 and 1.0 <= sh <= 11.0
    if passes < 21:
        assert int(100 * (100.0 * cl / total) + 0.5) == 50 and int(100 * (100.0 * ch / total) + 0.5) == 50
    again = oracle.stack_apply(2, frames, None, float(sl), float(sh))
    assert again[2:4] == (cl, ch) and np.array_equal(again[1], res, equal_nan=True)
