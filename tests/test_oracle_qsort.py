"""The reference's only test on the hot path, restated against the oracle:
internal/qsort/qsort_test.go:25-53 (TestMedian) -- median of a shuffled
permutation of 1..n for n = 1..999, exact float compare -- plus the other
entry points of internal/qsort/qsort.go."""
import numpy as np


def test_median_of_shuffled_permutations(oracle):
    # qsort_test.go:25-53; the reference shuffles with fastrand, any shuffle will do
    rng = np.random.default_rng(20260929)
    for n in range(1, 1000):
        arr = np.arange(1, n + 1, dtype=np.float32)
        for j in range(n):                      # same swap-with-random scheme as the test
            k = int(rng.integers(0, n))
            arr[j], arr[k] = arr[k], arr[j]
        if n & 1:
            expect = np.float32((n + 1) // 2)
        else:
            expect = np.float32(0.5) * (np.float32(n // 2) + np.float32(n // 2 + 1))
        got, _ = oracle.qselect_median(arr)
        assert got == expect, "median(1..%d) got %r expect %r" % (n, got, expect)


def test_qselect_returns_kth_smallest_and_partitions(oracle):
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 10, 57, 128, 513):
        a = rng.standard_normal(n).astype(np.float32)
        a[rng.integers(0, n, n // 4)] = a[0]            # ties
        s = np.sort(a)
        for k in {1, (n >> 2) + 1, (n >> 1) + 1, n}:
            v, perm = oracle.qselect(a, k)
            assert v == s[k - 1]
            assert np.array_equal(np.sort(perm), s)      # a permutation of the input
            assert perm[: k - 1].max(initial=-np.inf) <= v <= perm[k:].min(initial=np.inf)


def test_qsort_sorts(oracle):
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 5, 64, 1000):
        a = (rng.standard_normal(n) * 100).round().astype(np.float32)
        assert np.array_equal(oracle.qsort(a), np.sort(a))


def test_median_even_is_mean_of_middles(oracle):
    v, _ = oracle.qselect_median(np.array([4, 1, 3, 2], np.float32))
    assert v == np.float32(2.5)
    v, _ = oracle.qselect_median(np.array([7, 7, 7, 7], np.float32))
    assert v == np.float32(7)
