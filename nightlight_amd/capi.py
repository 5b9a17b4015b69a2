"""ctypes binding of the C ABI in include/nlstack.h (libnlstack.so).

Plumbing only: loads the in-tree HIP library and declares its prototypes.
There is no CPU fallback -- if the library is missing, or no HIP device is
visible, the compute entry points raise NlError.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# NLSTACK_LIB: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("NLSTACK_LIB") or os.path.join(_PKG, "libnlstack.so")

ST_MEDIAN, ST_MEAN, ST_SIGMA, ST_WINSOR_SIGMA, ST_MAD_SIGMA, ST_LINEAR_FIT, ST_AUTO = range(7)
WEIGHT_NONE, WEIGHT_EXPOSURE, WEIGHT_INVERSE_NOISE, WEIGHT_INVERSE_HFR = range(4)

OK = 0
ERR_INVALID_MODE = -1
ERR_MISSING_EXPOSURE = -2
ERR_INVALID_WEIGHTING = -3
ERR_WEIGHTED_MAD = -4
ERR_NO_INPUTS = -5
ERR_INVALID_ARG = -6
ERR_HIP = -7
ERR_TOO_MANY_FRAMES = -8
ERR_NO_DEVICE = -9

# every symbol include/nlstack.h declares (tests check the library exports them)
EXPORTS = [
    "nl_last_error", "nl_device_count", "nl_version",
    "nl_stack_create", "nl_stack_destroy",
    "nl_stack_upload_frame", "nl_stack_upload_tile", "nl_stack_upload_frame_async", "nl_stack_upload_wait", "nl_stack_frames_device_ptr", "nl_stack_device_bytes", "nl_release_cached_memory",
    "nl_fits_parse_header", "nl_fits_write_header", "nl_fits_padded_bytes",
    "nl_stack_attach_device_frames", "nl_stack_attach_device_frames_strided", "nl_stack_frame_stride", "nl_stack_fill_synthetic", "nl_stack_download_tile", "nl_stack_download_rows",
    "nl_stack_set_active_frames", "nl_stack_set_weights", "nl_weights_from_scalars",
    "nl_stack_linfit_stage_counts", "nl_stack_run", "nl_stack_run_async", "nl_stack_finish", "nl_stack_result_device_ptr",
    "nl_stack_last_mode", "nl_stack_last_kernel_ms", "nl_stack_last_dominant_kernel_ms",
    "nl_stack_last_kernel_name", "nl_stack_pass_times", "nl_stack_stream", "nl_stack_counters_device_ptr", "nl_stack_copy_counters_async", "nl_stack_set_counters_buffer", "nl_stack_order_stream_after",
    "nl_group_tile_rows", "nl_group_create", "nl_group_destroy", "nl_group_size", "nl_group_tile",
    "nl_group_upload_frame", "nl_group_fill_synthetic", "nl_group_set_active_frames", "nl_group_set_weights", "nl_group_set_exact",
    "nl_group_run", "nl_group_last_mode", "nl_group_find_sigmas", "nl_group_accumulate",
    "nl_group_accumulate_finalize",
    "nl_stack_set_exact", "nl_stack_set_dev_flags", "nl_stack_last_fallback_pixels", "nl_stack_last_generic_pixels", "nl_stack_last_pass_protocol",
    "nl_stack_find_sigmas", "nl_stack_accumulate", "nl_stack_accumulate_finalize",
    "nl_stack_frame_stats", "nl_stack_frame_noise", "nl_stack_weights_from_noise",
    "nl_median_filter_3x3", "nl_median_filter_mask",
    "nl_stack_upload_frame_fits", "nl_stack_upload_frame_projected", "nl_stack_frame_affine",
    "nl_stack_upload_frame_fits_async", "nl_stack_upload_frame_projected_async",
    "nl_group_upload_frame_fits", "nl_group_upload_frame_projected",
    "nl_stack_download_result_fits", "nl_fits_decode", "nl_project_bilinear",
    "nl_host_op_stack_apply_json", "nl_host_op_stack_roundtrip_json", "nl_host_set_devices",
    "nl_host_op_stack_batches_apply_json",
]


class NlError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("nlstack error %d: %s" % (code, message))
        self.code = code
        self.message = message


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int64), C.c_void_p)

_lib = None
_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_intp = C.POINTER(C.c_int)


class FitsHeader(C.Structure):
    """nl_fits_header_t (include/nlstack.h)"""
    _fields_ = [("bitpix", C.c_int32), ("naxis", C.c_int32), ("naxisn", C.c_int32 * 8),
                ("bzero", C.c_float), ("bscale", C.c_float), ("exposure", C.c_float),
                ("pixels", C.c_int64), ("header_bytes", C.c_int64), ("payload_bytes", C.c_int64),
                ("padded_payload_bytes", C.c_int64)]


def load():
    """Load libnlstack.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NlError(ERR_NO_DEVICE, "libnlstack.so is not built (%s); run "
                      "`python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    try:  # if torch is used in this process, let its bundled HIP runtime load first
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.nl_last_error.restype = C.c_char_p
    L.nl_version.restype = C.c_char_p
    L.nl_device_count.restype = C.c_int
    L.nl_stack_create.argtypes = [C.c_int] * 6
    L.nl_stack_create.restype = vp
    L.nl_stack_destroy.argtypes = [vp]
    L.nl_stack_destroy.restype = None
    L.nl_stack_upload_frame.argtypes = [vp, C.c_int, _f32p]
    L.nl_stack_upload_tile.argtypes = [vp, C.c_int, _f32p]
    L.nl_stack_download_tile.argtypes = [vp, C.c_int, _f32p]
    L.nl_stack_download_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, _f32p]
    L.nl_stack_frames_device_ptr.argtypes = [vp]
    L.nl_stack_frames_device_ptr.restype = vp
    L.nl_stack_device_bytes.argtypes = [vp]
    L.nl_stack_device_bytes.restype = C.c_int64
    L.nl_fits_parse_header.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(FitsHeader)]
    L.nl_fits_write_header.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_float, C.c_float, C.c_float]
    L.nl_fits_write_header.restype = C.c_int64
    L.nl_fits_padded_bytes.argtypes = [C.c_int64]
    L.nl_fits_padded_bytes.restype = C.c_int64
    L.nl_stack_attach_device_frames.argtypes = [vp, vp]
    L.nl_stack_attach_device_frames_strided.argtypes = [vp, vp, C.c_int64]
    L.nl_stack_frame_stride.argtypes = [vp]
    L.nl_stack_frame_stride.restype = C.c_int64
    L.nl_stack_fill_synthetic.argtypes = [vp, C.c_uint64]
    L.nl_stack_set_weights.argtypes = [vp, _f32p]
    L.nl_stack_set_active_frames.argtypes = [vp, C.c_int]
    L.nl_group_set_active_frames.argtypes = [vp, C.c_int]
    L.nl_weights_from_scalars.argtypes = [C.c_int, _f32p, C.c_int, _f32p, _intp]
    L.nl_stack_run.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _i64p, _i64p]
    L.nl_stack_run_async.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float]
    L.nl_stack_finish.argtypes = [vp, _f32p, _i64p, _i64p]
    L.nl_stack_result_device_ptr.argtypes = [vp]
    L.nl_stack_result_device_ptr.restype = vp
    L.nl_stack_last_mode.argtypes = [vp]
    L.nl_stack_last_kernel_ms.argtypes = [vp]
    L.nl_stack_last_kernel_ms.restype = C.c_float
    L.nl_stack_last_dominant_kernel_ms.argtypes = [vp]
    L.nl_stack_last_dominant_kernel_ms.restype = C.c_float
    L.nl_stack_last_kernel_name.argtypes = [vp]
    L.nl_stack_last_kernel_name.restype = C.c_char_p
    L.nl_stack_set_exact.argtypes = [vp, C.c_int]
    L.nl_stack_set_dev_flags.argtypes = [vp, C.c_uint]
    L.nl_stack_pass_times.argtypes = [vp, C.c_int, _f32p, _f32p]
    L.nl_stack_stream.argtypes = [vp]
    L.nl_stack_stream.restype = vp
    L.nl_stack_counters_device_ptr.argtypes = [vp]
    L.nl_stack_counters_device_ptr.restype = vp
    L.nl_stack_copy_counters_async.argtypes = [vp, vp]
    L.nl_stack_set_counters_buffer.argtypes = [vp, vp]
    L.nl_stack_order_stream_after.argtypes = [vp, vp]
    L.nl_stack_set_counters_buffer.restype = C.c_int
    L.nl_group_tile_rows.argtypes = [C.c_int, C.c_int, C.c_int, _intp, _intp]
    L.nl_group_tile_rows.restype = None
    L.nl_group_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _intp]
    L.nl_group_create.restype = vp
    L.nl_group_destroy.argtypes = [vp]
    L.nl_group_destroy.restype = None
    L.nl_group_size.argtypes = [vp]
    L.nl_group_tile.argtypes = [vp, C.c_int]
    L.nl_group_tile.restype = vp
    L.nl_group_upload_frame.argtypes = [vp, C.c_int, _f32p]
    L.nl_group_fill_synthetic.argtypes = [vp, C.c_uint64]
    L.nl_group_set_weights.argtypes = [vp, _f32p]
    L.nl_group_set_exact.argtypes = [vp, C.c_int]
    L.nl_group_run.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _i64p, _i64p]
    L.nl_group_last_mode.argtypes = [vp]
    L.nl_group_find_sigmas.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _i64p, _i64p,
                                       _f32p, _f32p, _intp]
    L.nl_group_accumulate.argtypes = [vp, C.c_float, C.c_int]
    L.nl_group_accumulate_finalize.argtypes = [vp, C.c_float, _f32p]
    L.nl_stack_last_fallback_pixels.argtypes = [vp]
    L.nl_stack_last_fallback_pixels.restype = C.c_int64
    L.nl_stack_last_generic_pixels.argtypes = [vp]
    L.nl_stack_last_pass_protocol.argtypes = [vp]
    L.nl_stack_last_generic_pixels.restype = C.c_int64
    L.nl_stack_linfit_stage_counts.argtypes = [vp, C.POINTER(C.c_uint), C.c_int]
    L.nl_stack_linfit_stage_counts.restype = C.c_int
    L.nl_stack_find_sigmas.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, REDUCE_FN,
                                       vp, _f32p, _i64p, _i64p, _f32p, _f32p, _intp]
    L.nl_stack_accumulate.argtypes = [vp, C.c_float, C.c_int]
    L.nl_stack_accumulate_finalize.argtypes = [vp, C.c_float, _f32p]
    L.nl_stack_frame_stats.argtypes = [vp, C.c_int, _f32p, _f32p, _f32p, C.POINTER(C.c_double)]
    L.nl_stack_frame_noise.argtypes = [vp, C.c_int, _f32p]
    L.nl_stack_weights_from_noise.argtypes = [vp, _f32p]
    L.nl_median_filter_3x3.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int]
    L.nl_median_filter_mask.argtypes = [_f32p, _f32p, C.c_int64, C.POINTER(C.c_int32), C.c_int, C.c_int]
    L.nl_stack_upload_frame_async.argtypes = [vp, C.c_int, _f32p]
    L.nl_stack_upload_wait.argtypes = [vp]
    L.nl_stack_upload_frame_fits.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_float,
                                             C.c_float, _f32p]
    L.nl_stack_upload_frame_fits_async.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    L.nl_group_upload_frame_fits.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    L.nl_stack_upload_frame_projected_async.argtypes = [vp, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_float,
                                                          C.c_float, C.c_float]
    L.nl_group_upload_frame_projected.argtypes = [vp, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_float,
                                                   C.c_float, C.c_float]
    L.nl_stack_upload_frame_projected.argtypes = [vp, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_float,
                                                  C.c_float, C.c_float]
    L.nl_stack_frame_affine.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    L.nl_stack_download_result_fits.argtypes = [vp, vp]
    L.nl_fits_decode.argtypes = [vp, C.c_int, C.c_int64, C.c_float, C.c_float, _f32p, _f32p, C.c_int]
    L.nl_project_bilinear.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_float,
                                      C.c_int]
    L.nl_host_op_stack_apply_json.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(_f32p), _f32p, _f32p, C.c_int, C.c_int,
                                              _f32p, _f32p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.nl_host_op_stack_batches_apply_json.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_f32p),
                                                      _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p,
                                                      _intp, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.nl_host_set_devices.argtypes = [_intp, C.c_int]
    L.nl_host_op_stack_roundtrip_json.argtypes = [C.c_char_p]
    L.nl_host_op_stack_roundtrip_json.restype = C.c_char_p
    _lib = L
    return L


def last_error():
    return load().nl_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        raise NlError(rc, last_error())
    return rc


def fptr(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_f32p)


def device_count():
    n = load().nl_device_count()
    return n if n > 0 else 0
