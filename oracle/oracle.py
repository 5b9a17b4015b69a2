"""ctypes loader for the CPU oracle (oracle/libnl_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package nightlight_amd.
Each wrapper mirrors one function of oracle/nl_oracle.h, which cites the
reference file:line it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnl_oracle.so")

ST_MEDIAN, ST_MEAN, ST_SIGMA, ST_WINSOR_SIGMA, ST_MAD_SIGMA, ST_LINEAR_FIT, ST_AUTO = range(7)
WEIGHT_NONE, WEIGHT_EXPOSURE, WEIGHT_INVERSE_NOISE, WEIGHT_INVERSE_HFR = range(4)

OK = 0
ERR_INVALID_MODE = -1
ERR_MISSING_EXPOSURE = -2
ERR_INVALID_WEIGHTING = -3
ERR_WEIGHTED_MAD = -4
ERR_NO_INPUTS = -5


def build(force=False):
    """Compile the oracle with its Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "nl_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None
_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_pp = C.POINTER(_f32p)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.nlo_qsort_f32.argtypes = [_f32p, C.c_int]
        L.nlo_qsort_f32.restype = None
        L.nlo_qpartition_f32.argtypes = [_f32p, C.c_int]
        L.nlo_qpartition_f32.restype = C.c_int
        L.nlo_qselect_f32.argtypes = [_f32p, C.c_int, C.c_int]
        L.nlo_qselect_f32.restype = C.c_float
        L.nlo_qselect_median_f32.argtypes = [_f32p, C.c_int]
        L.nlo_qselect_median_f32.restype = C.c_float
        L.nlo_qselect_first_quartile_f32.argtypes = [_f32p, C.c_int]
        L.nlo_qselect_first_quartile_f32.restype = C.c_float
        L.nlo_mean_stddev.argtypes = [_f32p, C.c_int, _f32p, _f32p]
        L.nlo_mean_stddev.restype = None
        L.nlo_linear_regression.argtypes = [_f32p, _f32p, C.c_int] + [_f32p] * 6
        L.nlo_linear_regression.restype = None
        for name in ("nlo_min_mean_max", "nlo_min_mean_max_lanes4"):
            f = getattr(L, name)
            f.argtypes = [_f32p, C.c_int64, _f32p, _f32p, _f32p]
            f.restype = None
        for name in ("nlo_variance", "nlo_variance_lanes4"):
            f = getattr(L, name)
            f.argtypes = [_f32p, C.c_int64, C.c_float]
            f.restype = C.c_double
        L.nlo_estimate_noise.argtypes = [_f32p, C.c_int64, C.c_int32]
        L.nlo_estimate_noise.restype = C.c_float
        L.nlo_median9.argtypes = [_f32p]
        L.nlo_median9.restype = C.c_float
        L.nlo_median_f32.argtypes = [_f32p, C.c_int]
        L.nlo_median_f32.restype = C.c_float
        L.nlo_median_filter_3x3.argtypes = [_f32p, _f32p, C.c_int64, C.c_int32]
        L.nlo_median_filter_3x3.restype = None
        L.nlo_auto_select_mode.argtypes = [C.c_int]
        L.nlo_auto_select_mode.restype = C.c_int
        L.nlo_get_weights.argtypes = [C.c_int, _f32p, C.c_int, _f32p,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.nlo_get_weights.restype = C.c_int
        L.nlo_stack_apply.argtypes = [C.c_int, _pp, _f32p, C.c_int, C.c_int64, C.c_float,
                                      C.c_float, C.c_float, C.c_int, _f32p, _i64p, _i64p,
                                      C.POINTER(C.c_int)]
        L.nlo_stack_apply.restype = C.c_int
        L.nlo_stack_incremental.argtypes = [_f32p, _f32p, C.c_int64, C.c_float, C.c_int]
        L.nlo_stack_incremental.restype = None
        L.nlo_stack_incremental_finalize.argtypes = [_f32p, C.c_int64, C.c_float]
        L.nlo_stack_incremental_finalize.restype = None
        L.nlo_find_sigmas_bisect.argtypes = [C.c_int, _pp, _f32p, C.c_int, C.c_int64, C.c_float,
                                             C.c_float, C.c_float, C.c_int, _f32p, _i64p, _i64p,
                                             _f32p, _f32p]
        L.nlo_find_sigmas_bisect.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _frame_ptrs(frames):
    """frames: [N, P] array or list of N 1-D arrays -> (keepalive, float** )."""
    if isinstance(frames, np.ndarray):
        assert frames.dtype == np.float32 and frames.flags.c_contiguous
        rows = [frames[i].reshape(-1) for i in range(frames.shape[0])]
    else:
        rows = [_f32(f).reshape(-1) for f in frames]
    arr = (_f32p * len(rows))(*[_fp(r) for r in rows])
    return rows, arr


# ---- qsort.go -------------------------------------------------------------
def qsort(a):
    a = _f32(a).copy()
    lib().nlo_qsort_f32(_fp(a), a.size)
    return a


def qselect(a, k):
    """returns (value, permuted array) -- k is 1-based like the reference."""
    a = _f32(a).copy()
    v = lib().nlo_qselect_f32(_fp(a), a.size, int(k))
    return np.float32(v), a


def qselect_median(a):
    a = _f32(a).copy()
    v = lib().nlo_qselect_median_f32(_fp(a), a.size)
    return np.float32(v), a


# ---- stats.go ---------------------------------------------------------------
def mean_stddev(xs):
    xs = _f32(xs)
    m, s = C.c_float(), C.c_float()
    lib().nlo_mean_stddev(_fp(xs), xs.size, C.byref(m), C.byref(s))
    return np.float32(m.value), np.float32(s.value)


def linear_regression(xs, ys):
    xs, ys = _f32(xs), _f32(ys)
    o = [C.c_float() for _ in range(6)]
    lib().nlo_linear_regression(_fp(xs), _fp(ys), xs.size, *[C.byref(x) for x in o])
    return tuple(np.float32(x.value) for x in o)  # slope, intercept, xmean, xstd, ymean, ystd


def min_mean_max(data, lanes4=False):
    data = _f32(data).reshape(-1)
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    f = lib().nlo_min_mean_max_lanes4 if lanes4 else lib().nlo_min_mean_max
    f(_fp(data), data.size, C.byref(a), C.byref(b), C.byref(c))
    return np.float32(a.value), np.float32(b.value), np.float32(c.value)


def variance(data, mean, lanes4=False):
    data = _f32(data).reshape(-1)
    f = lib().nlo_variance_lanes4 if lanes4 else lib().nlo_variance
    return float(f(_fp(data), data.size, C.c_float(float(mean))))


def estimate_noise(data, width):
    data = _f32(data).reshape(-1)
    return np.float32(lib().nlo_estimate_noise(_fp(data), data.size, int(width)))


def median9(a):
    a = _f32(a).copy()
    assert a.size == 9
    return np.float32(lib().nlo_median9(_fp(a)))


def median_filter_3x3(data, width):
    data = _f32(data).reshape(-1)
    out = np.empty_like(data)
    lib().nlo_median_filter_3x3(_fp(out), _fp(data), data.size, int(width))
    return out


# ---- stack.go ---------------------------------------------------------------
def auto_select_mode(n):
    return lib().nlo_auto_select_mode(int(n))


def get_weights(weighting, per_frame):
    per_frame = _f32(per_frame)
    w = np.zeros(per_frame.size, np.float32)
    has, bad = C.c_int(), C.c_int()
    rc = lib().nlo_get_weights(int(weighting), _fp(per_frame), per_frame.size, _fp(w),
                               C.byref(has), C.byref(bad))
    return rc, (w if has.value else None), bad.value


def stack_apply(mode, frames, weights=None, sigma_low=2.75, sigma_high=2.75,
                ref_loc=0.0, num_cpu=1):
    """OpStack.Apply numeric core. frames: [N, P] float32.
    Returns (rc, result[P], clip_low, clip_high, mode_used)."""
    keep, ptrs = _frame_ptrs(frames)
    n = len(keep)
    npix = keep[0].size if n else 0
    res = np.empty(npix, np.float32)
    cl, ch, mu = C.c_int64(0), C.c_int64(0), C.c_int(-1)
    wp = None
    if weights is not None:
        weights = _f32(weights)
        wp = _fp(weights)
    rc = lib().nlo_stack_apply(int(mode), ptrs, wp, n, npix, C.c_float(ref_loc),
                               C.c_float(sigma_low), C.c_float(sigma_high), int(num_cpu),
                               _fp(res), C.byref(cl), C.byref(ch), C.byref(mu))
    return rc, res, cl.value, ch.value, mu.value


def stack_incremental(stack, light, weight, first):
    stack = _f32(stack).reshape(-1)
    light = _f32(light).reshape(-1)
    lib().nlo_stack_incremental(_fp(stack), _fp(light), stack.size, C.c_float(weight), int(first))
    return stack


def stack_incremental_finalize(stack, weight_sum):
    stack = _f32(stack).reshape(-1)
    lib().nlo_stack_incremental_finalize(_fp(stack), stack.size, C.c_float(weight_sum))
    return stack


def find_sigmas_bisect(mode, frames, clip_perc_low, clip_perc_high, weights=None,
                       ref_loc=0.0, num_cpu=1):
    keep, ptrs = _frame_ptrs(frames)
    n = len(keep)
    npix = keep[0].size
    res = np.empty(npix, np.float32)
    cl, ch = C.c_int64(0), C.c_int64(0)
    sl, sh = C.c_float(), C.c_float()
    wp = None
    if weights is not None:
        weights = _f32(weights)
        wp = _fp(weights)
    passes = lib().nlo_find_sigmas_bisect(int(mode), ptrs, wp, n, npix, C.c_float(ref_loc),
                                          C.c_float(clip_perc_low), C.c_float(clip_perc_high),
                                          int(num_cpu), _fp(res), C.byref(cl), C.byref(ch),
                                          C.byref(sl), C.byref(sh))
    return passes, res, cl.value, ch.value, np.float32(sl.value), np.float32(sh.value)
