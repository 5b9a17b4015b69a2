"""Whole-frame statistics of the oracle (internal/stats, internal/median)
against straightforward numpy formulations."""
import numpy as np


def test_min_mean_max_and_variance(oracle):
    rng = np.random.default_rng(1)
    d = (1000 + 30 * rng.standard_normal(64 * 48)).astype(np.float32)
    for lanes4 in (False, True):
        mn, mean, mx = oracle.min_mean_max(d, lanes4=lanes4)
        assert mn == d.min() and mx == d.max()
        assert mean == np.float32(d.astype(np.float64).sum() / d.size)
        var = oracle.variance(d, mean, lanes4=lanes4)
        want = (((d - mean).astype(np.float64)) ** 2).sum() / d.size
        assert abs(var - want) <= 1e-12 * want


def test_estimate_noise_matches_direct_convolution(oracle):
    rng = np.random.default_rng(2)
    w, h = 40, 25
    img = (500 + 20 * rng.standard_normal((h, w))).astype(np.float32)
    k = np.array([[1, -2, 1], [-2, 4, -2], [1, -2, 1]], np.float64)
    acc = 0.0
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            acc += abs((img[y - 1:y + 2, x - 1:x + 2].astype(np.float64) * k).sum())
    want = acc * np.sqrt(0.5 * np.pi) / (6.0 * (w - 2) * (h - 2))
    got = oracle.estimate_noise(img, w)
    assert abs(got - want) <= 2e-5 * want
    assert abs(got / 20.0 - 1) < 0.15           # it estimates the noise sigma


def test_median9_network_is_the_median(oracle):
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = rng.integers(0, 6, 9).astype(np.float32)      # many ties
        assert oracle.median9(a) == np.median(a)


def test_median_filter_3x3(oracle):
    rng = np.random.default_rng(4)
    w, h = 23, 17
    img = rng.standard_normal((h, w)).astype(np.float32)
    out = oracle.median_filter_3x3(img, w).reshape(h, w)
    want = img.copy()
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            want[y, x] = np.median(img[y - 1:y + 2, x - 1:x + 2])
    assert np.array_equal(out, want)


def test_gather_and_median_reads_the_whole_buffer(oracle):
    # gather.go:37: MedianFloat32(buffer), not buffer[:num] -- at the data's edges the result depends on
    # what the previous call left behind; in the interior it is the plain median of the neighbourhood
    import numpy as np
    width, height = 32, 9
    rng = np.random.default_rng(4)
    img = rng.normal(100, 10, width * height).astype(np.float32)
    mask = oracle.create_mask(width, 1.5)
    assert list(mask) == [-33, -32, -31, -1, 0, 1, 31, 32, 33]
    out, full = oracle.median_filter_mask(img, mask)
    g = img.reshape(height, width)
    nine = np.stack([g[dy:height - 2 + dy, dx:width - 2 + dx] for dy in range(3) for dx in range(3)])
    assert np.array_equal(out.reshape(height, width)[1:-1, 1:-1], np.median(nine, axis=0))
    # first pixel: offsets 0, 1, 31, 32, 33 exist (the mask works on the LINEAR index, so +31 is the end
    # of row 0), the other 4 buffer slots are still the calloc'd zeros -> the 5th smallest of the 9
    assert out[0] == min(img[0], img[1], img[31], img[32], img[33])
    assert not full[0] and full[33] and not full[32]        # x = 0 of an inner row wraps to the previous row: still inside
    assert oracle.median_f32([]) != oracle.median_f32([])       # len 0 -> NaN (median3x3.go:116)
    assert oracle.median_f32([4, 1, 3, 2]) == 2.5


def test_newton_goal_seek_keeps_the_reference_quirks(oracle):
    # stackfindsigma.go:101-170: starts at (6, 6); both probes move a sigma by 0.005 -- on a small
    # stack no counter changes, the derivative is 0 and the search stops after 2 passes with (6, 6)
    import numpy as np
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import make_frames
    frames = make_frames(30, 64, 16, seed=5)
    passes, res, cl, ch, sl, sh = oracle.find_sigmas_newton(5, frames, 1.0, 1.0, num_cpu=2)
    rc, want, wl, wh, _ = oracle.stack_apply(5, frames, None, 6.0, 6.0)
    assert (passes, cl, ch, float(sl), float(sh)) == (2, wl, wh, 6.0, 6.0)
    assert np.array_equal(res, want, equal_nan=True)
