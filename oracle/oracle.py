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
        _i32p = C.POINTER(C.c_int32)
        L.nlo_median_filter_mask.argtypes = [_f32p, _f32p, C.c_int64, _i32p, C.c_int, C.POINTER(C.c_ubyte)]
        L.nlo_median_filter_mask.restype = None
        L.nlo_create_mask.argtypes = [C.c_int32, C.c_float, _i32p, C.c_int]
        L.nlo_create_mask.restype = C.c_int
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
        L.nlo_set_pin_workers.argtypes = [C.c_int]
        L.nlo_set_pin_workers.restype = None
        _u8p = C.POINTER(C.c_ubyte)
        L.nlo_fits_decode.argtypes = [_u8p, C.c_int, C.c_int64, C.c_float, C.c_float, _f32p, _f32p, _f32p, _f32p]
        L.nlo_fits_decode.restype = C.c_int
        L.nlo_fits_encode.argtypes = [_f32p, C.c_int64, C.c_int, _u8p]
        L.nlo_fits_encode.restype = None
        L.nlo_affine.argtypes = [_f32p, C.c_int64, C.c_float, C.c_float]
        L.nlo_affine.restype = None
        L.nlo_transform_invert.argtypes = [_f32p, _f32p]
        L.nlo_transform_invert.restype = C.c_int
        L.nlo_project_bilinear.argtypes = [_f32p, C.c_int32, C.c_int32, _f32p, C.c_int32, C.c_int32, _f32p, C.c_float]
        L.nlo_project_bilinear.restype = C.c_int
        L.nlo_stack_incremental.argtypes = [_f32p, _f32p, C.c_int64, C.c_float, C.c_int]
        L.nlo_stack_incremental.restype = None
        L.nlo_stack_incremental_finalize.argtypes = [_f32p, C.c_int64, C.c_float]
        L.nlo_stack_incremental_finalize.restype = None
        L.nlo_find_sigmas_bisect.argtypes = [C.c_int, _pp, _f32p, C.c_int, C.c_int64, C.c_float,
                                             C.c_float, C.c_float, C.c_int, _f32p, _i64p, _i64p,
                                             _f32p, _f32p]
        L.nlo_find_sigmas_bisect.restype = C.c_int
        L.nlo_find_sigmas_newton.argtypes = L.nlo_find_sigmas_bisect.argtypes
        L.nlo_find_sigmas_newton.restype = C.c_int
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


def median_f32(a):
    a = _f32(a).copy()
    return np.float32(lib().nlo_median_f32(_fp(a), a.size))


def create_mask(width, radius):
    """star/findstars.go:187-200."""
    m = np.zeros(256, np.int32)
    n = lib().nlo_create_mask(int(width), float(radius), m.ctypes.data_as(C.POINTER(C.c_int32)), m.size)
    return m[:n].copy()


def median_filter_mask(data, mask):
    """MedianFilter (badpixels.go:54-77) walked by one goroutine; returns (out, full) where full marks
    the pixels whose whole neighbourhood lies inside the data (result independent of call history)."""
    data = _f32(data).reshape(-1)
    mask = np.ascontiguousarray(mask, np.int32)
    out = np.empty_like(data)
    full = np.zeros(data.size, np.uint8)
    lib().nlo_median_filter_mask(_fp(out), _fp(data), data.size, mask.ctypes.data_as(C.POINTER(C.c_int32)),
                                 mask.size, full.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out, full.astype(bool)


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


def set_pin_workers(on):
    """Timing aid: pin the workers of stack_apply's pool to the allowed CPUs (results never depend on it)."""
    lib().nlo_set_pin_workers(1 if on else 0)


def stack_incremental(stack, light, weight, first):
    stack = _f32(stack).reshape(-1)
    light = _f32(light).reshape(-1)
    lib().nlo_stack_incremental(_fp(stack), _fp(light), stack.size, C.c_float(weight), int(first))
    return stack


def stack_incremental_finalize(stack, weight_sum):
    stack = _f32(stack).reshape(-1)
    lib().nlo_stack_incremental_finalize(_fp(stack), stack.size, C.c_float(weight_sum))
    return stack


def find_sigmas_newton(mode, frames, clip_perc_low, clip_perc_high, weights=None, ref_loc=0.0, num_cpu=1):
    """stackfindsigma.go:101-170; same return tuple as find_sigmas_bisect."""
    return find_sigmas_bisect(mode, frames, clip_perc_low, clip_perc_high, weights, ref_loc, num_cpu,
                              _fn="nlo_find_sigmas_newton")


def find_sigmas_bisect(mode, frames, clip_perc_low, clip_perc_high, weights=None,
                       ref_loc=0.0, num_cpu=1, _fn="nlo_find_sigmas_bisect"):
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
    passes = getattr(lib(), _fn)(int(mode), ptrs, wp, n, npix, C.c_float(ref_loc),
                                          C.c_float(clip_perc_low), C.c_float(clip_perc_high),
                                          int(num_cpu), _fp(res), C.byref(cl), C.byref(ch),
                                          C.byref(sl), C.byref(sh))
    return passes, res, cl.value, ch.value, np.float32(sl.value), np.float32(sh.value)


# ---- fits/read.go, fits/write.go, fits/pixelops.go, fits/project.go ---------------
BYTES_PER_VALUE = {8: 1, 16: 2, 32: 4, 64: 8, -32: 4, -64: 8}


def fits_decode(raw, bitpix, bscale=1.0, bzero=0.0):
    """raw: bytes / uint8 array of a big-endian FITS payload -> (rc, fp32 data, min, max, mean)."""
    raw = np.ascontiguousarray(np.frombuffer(bytes(raw), np.uint8) if not isinstance(raw, np.ndarray) else raw,
                               np.uint8)
    if bitpix not in BYTES_PER_VALUE:
        return -1, None, 0.0, 0.0, 0.0
    n = raw.size // BYTES_PER_VALUE[bitpix]
    out = np.empty(n, np.float32)
    mn, mx, mean = C.c_float(), C.c_float(), C.c_float()
    rc = lib().nlo_fits_decode(raw.ctypes.data_as(C.POINTER(C.c_ubyte)), int(bitpix), n, float(bscale),
                               float(bzero), _fp(out), C.byref(mn), C.byref(mx), C.byref(mean))
    return rc, out, mn.value, mx.value, mean.value


def fits_encode(data, replace_nans=True):
    data = _f32(data).reshape(-1)
    raw = np.empty(data.size * 4, np.uint8)
    lib().nlo_fits_encode(_fp(data), data.size, int(bool(replace_nans)), raw.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return raw


def affine(data, multiplier, offset):
    out = _f32(data).reshape(-1).copy()
    lib().nlo_affine(_fp(out), out.size, float(multiplier), float(offset))
    return out


def transform_invert(trans):
    t = _f32(trans).reshape(6)
    inv = np.zeros(6, np.float32)
    rc = lib().nlo_transform_invert(_fp(t), _fp(inv))
    return rc, inv


def project_bilinear(src, src_w, src_h, dst_w, dst_h, trans, out_of_bounds=np.nan):
    src = _f32(src).reshape(-1)
    t = _f32(trans).reshape(6)
    dst = np.empty(int(dst_w) * int(dst_h), np.float32)
    rc = lib().nlo_project_bilinear(_fp(src), int(src_w), int(src_h), _fp(dst), int(dst_w), int(dst_h),
                                    _fp(t), float(out_of_bounds))
    return rc, dst
