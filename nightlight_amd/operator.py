"""Python view of the host-side operator mirror (nightlight_amd/host/*.cpp):
the reference's "stack" operator decoded from JSON and run through
MakePromises/Apply (internal/ops/stack/stack.go:92-227)."""
import ctypes as C

import numpy as np

from . import capi


class OperatorError(RuntimeError):
    pass


def op_stack_roundtrip_json(text):
    """UnmarshalJSON with the reference's defaults, then MarshalJSON."""
    return capi.load().nl_host_op_stack_roundtrip_json(text.encode()).decode()


def op_stack_apply_json(text, frames, width, height, exposure=None, hfr=None, device=0,
                        max_threads=4):
    """frames: list of full-image float32 arrays, None = frame skipped upstream.
    Returns (result, exposure_sum, log)."""
    lib = capi.load()
    f32p = C.POINTER(C.c_float)
    keep = [None if f is None else np.ascontiguousarray(f, np.float32).reshape(-1) for f in frames]
    ptrs = (f32p * max(len(keep), 1))(*[None if k is None else k.ctypes.data_as(f32p) for k in keep])
    out = np.zeros(width * height, np.float32)
    exp_out = C.c_float(0)
    log = C.create_string_buffer(4096)
    err = C.create_string_buffer(1024)
    e = None if exposure is None else np.ascontiguousarray(exposure, np.float32)
    h = None if hfr is None else np.ascontiguousarray(hfr, np.float32)
    rc = lib.nl_host_op_stack_apply_json(text.encode(), len(keep), int(width), int(height), ptrs,
                                         None if e is None else capi.fptr(e),
                                         None if h is None else capi.fptr(h),
                                         int(device), int(max_threads), capi.fptr(out),
                                         C.byref(exp_out), log, len(log), err, len(err))
    if rc != 0:
        raise OperatorError(err.value.decode())
    return out, float(exp_out.value), log.value.decode()


def set_devices(devices):
    """Devices every host-side operator of this process stacks on (one row tile of each stack
    per entry; a device may repeat).  None / empty: back to the `device` argument."""
    lib = capi.load()
    d = list(devices or [])
    arr = (C.c_int * max(len(d), 1))(*d)
    lib.nl_host_set_devices(arr, len(d))


def op_stack_batches_apply_json(per_batch_json, frames, width, height, exposure=None, device=0,
                                max_threads=4, memory_mb=0, stack_memory_mb=0, return_perm=False):
    """OpStackBatches (stackbatches.go:46-217) over host frames.
    Returns (result, exposure_sum, log[, perm]); raises OperatorError with the reference's message.
    perm[i] = input index of position i after partitioning; batches are consecutive runs."""
    lib = capi.load()
    f32p = C.POINTER(C.c_float)
    keep = [np.ascontiguousarray(f, np.float32).reshape(-1) for f in frames]
    ptrs = (f32p * max(len(keep), 1))(*[k.ctypes.data_as(f32p) for k in keep])
    out = np.zeros(width * height, np.float32)
    exp_out = C.c_float(0)
    log = C.create_string_buffer(16384)
    err = C.create_string_buffer(1024)
    perm = (C.c_int * max(len(keep), 1))()
    e = None if exposure is None else np.ascontiguousarray(exposure, np.float32)
    rc = lib.nl_host_op_stack_batches_apply_json(
        per_batch_json.encode(), len(keep), int(width), int(height), ptrs,
        None if e is None else capi.fptr(e), int(device), int(max_threads), int(memory_mb),
        int(stack_memory_mb), capi.fptr(out), C.byref(exp_out), perm, log, len(log), err, len(err))
    if rc != 0:
        raise OperatorError(err.value.decode() + ("\n" + log.value.decode() if log.value else ""))
    if return_perm:
        return out, float(exp_out.value), log.value.decode(), list(perm)[:len(keep)]
    return out, float(exp_out.value), log.value.decode()
