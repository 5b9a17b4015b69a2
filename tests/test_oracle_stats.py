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
