"""Thin Python view of one nl_stack_t handle (include/nlstack.h).

Used by the tests, bench.py and the row-tile sharding helper; every method is
one C-ABI call.  Naming follows the reference's domain: frames, tiles, stack
passes, clip counters (internal/ops/stack/stack.go).
"""
import ctypes as C

import numpy as np

from . import capi


class StackHandle:
    """Frames of one row tile [row0, row0+rows) resident in HBM as planar
    [n_frames][rows*width] fp32, plus the result tile and clip counters."""

    def __init__(self, n_frames, width, height, row0=0, rows=None, device=0):
        self._lib = capi.load()
        rows = height - row0 if rows is None else rows
        self.n_frames, self.width, self.height = int(n_frames), int(width), int(height)
        self.row0, self.rows, self.device = int(row0), int(rows), int(device)
        self._h = self._lib.nl_stack_create(self.n_frames, self.width, self.height,
                                            self.row0, self.rows, self.device)
        if not self._h:
            raise capi.NlError(capi.ERR_HIP, capi.last_error())

    @classmethod
    def _borrow(cls, handle, n_frames, width, height, row0, rows):
        """View of a handle owned by someone else (a tile of an nl_group): never destroyed here."""
        self = cls.__new__(cls)
        self._lib = capi.load()
        self.n_frames, self.width, self.height = int(n_frames), int(width), int(height)
        self.row0, self.rows, self.device = int(row0), int(rows), -1
        self._h, self._borrowed = handle, True
        return self

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if self._h:
            if not getattr(self, "_borrowed", False):
                self._lib.nl_stack_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def tile_pixels(self):
        return self.rows * self.width

    # -- frames ------------------------------------------------------------
    def upload_frame(self, idx, frame):
        """frame: full image, width*height float32 (fits.Image.Data)."""
        frame = np.ascontiguousarray(frame, dtype=np.float32).reshape(-1)
        assert frame.size == self.width * self.height
        capi.check(self._lib.nl_stack_upload_frame(self._h, int(idx), capi.fptr(frame)))

    def upload_frame_async(self, idx, frame):
        """Overlapped upload through the pinned staging ring (frame = whole image)."""
        frame = np.ascontiguousarray(frame, dtype=np.float32).reshape(-1)
        assert frame.size == self.width * self.height
        capi.check(self._lib.nl_stack_upload_frame_async(self._h, int(idx), capi.fptr(frame)))

    def upload_wait(self):
        capi.check(self._lib.nl_stack_upload_wait(self._h))

    def upload_tile(self, idx, tile):
        tile = np.ascontiguousarray(tile, dtype=np.float32).reshape(-1)
        assert tile.size == self.tile_pixels
        capi.check(self._lib.nl_stack_upload_tile(self._h, int(idx), capi.fptr(tile)))

    def upload_frames(self, frames):
        for i, f in enumerate(frames):
            self.upload_frame(i, f)

    def download_tile(self, idx):
        out = np.empty(self.tile_pixels, np.float32)
        capi.check(self._lib.nl_stack_download_tile(self._h, int(idx), capi.fptr(out)))
        return out

    def download_rows(self, idx, first_row, n_rows):
        """Rows [first_row, first_row+n_rows) (tile-relative) of frame idx; idx=-1: of the result."""
        out = np.empty(int(n_rows) * self.width, np.float32)
        capi.check(self._lib.nl_stack_download_rows(self._h, int(idx), int(first_row), int(n_rows),
                                                    capi.fptr(out)))
        return out

    def fill_synthetic(self, seed=0x4E4C5354):
        capi.check(self._lib.nl_stack_fill_synthetic(self._h, C.c_uint64(seed)))

    @property
    def device_bytes(self):
        """Device memory the handle holds right now (create-time buffers + lazily allocated scratch)."""
        return int(self._lib.nl_stack_device_bytes(self._h))

    def frames_device_ptr(self):
        return self._lib.nl_stack_frames_device_ptr(self._h)

    def frame_stride(self):
        """Floats between consecutive frames of the buffer frames_device_ptr() points at."""
        return int(self._lib.nl_stack_frame_stride(self._h))

    def attach_device_frames(self, ptr, stride=None):
        """Lends a device buffer (None restores the owned one); `stride` in floats, default dense (rows*width)."""
        if stride is None or ptr is None:
            capi.check(self._lib.nl_stack_attach_device_frames(self._h, C.c_void_p(ptr)))
        else:
            capi.check(self._lib.nl_stack_attach_device_frames_strided(self._h, C.c_void_p(ptr), int(stride)))

    def set_active_frames(self, n):
        """Use frame slots [0, n) for the next uploads / passes (n <= the count given at creation)."""
        capi.check(self._lib.nl_stack_set_active_frames(self._h, int(n)))
        self.n_frames = int(n)

    def set_weights(self, weights):
        if weights is None:
            capi.check(self._lib.nl_stack_set_weights(self._h, None))
            return
        w = np.ascontiguousarray(weights, dtype=np.float32)
        assert w.size == self.n_frames
        capi.check(self._lib.nl_stack_set_weights(self._h, capi.fptr(w)))

    # -- stack passes ------------------------------------------------------
    def run(self, mode, sigma_low=2.75, sigma_high=2.75, ref_loc=0.0, out=None, fetch=True):
        """One pass. Returns (result or None, clip_low, clip_high).  `out` is a
        full-image float32 array whose tile rows get written; with fetch=False
        the result stays on the device."""
        cl, ch = C.c_int64(0), C.c_int64(0)
        if fetch and out is None:
            out = np.zeros(self.width * self.height, np.float32)
        optr = capi.fptr(out) if (fetch and out is not None) else None
        capi.check(self._lib.nl_stack_run(self._h, int(mode), C.c_float(sigma_low),
                                          C.c_float(sigma_high), C.c_float(ref_loc), optr,
                                          C.byref(cl), C.byref(ch)))
        return (out if fetch else None), cl.value, ch.value

    def run_async(self, mode, sigma_low=2.75, sigma_high=2.75, ref_loc=0.0):
        capi.check(self._lib.nl_stack_run_async(self._h, int(mode), C.c_float(sigma_low),
                                                C.c_float(sigma_high), C.c_float(ref_loc)))

    def finish(self, out=None):
        cl, ch = C.c_int64(0), C.c_int64(0)
        capi.check(self._lib.nl_stack_finish(self._h, capi.fptr(out) if out is not None else None,
                                             C.byref(cl), C.byref(ch)))
        return cl.value, ch.value

    def result_tile(self):
        """The result tile only (rows*width), downloaded from the device."""
        full = np.zeros(self.width * self.height, np.float32)
        self.finish(full)
        return full[self.row0 * self.width:(self.row0 + self.rows) * self.width].copy()

    def set_exact(self, on=True):
        """Force the bit-exact kernels (verification); default is the fast path."""
        capi.check(self._lib.nl_stack_set_exact(self._h, int(on)))

    def set_dev_flags(self, flags):
        """A/B switches of the fast path (include/nlstack.h: nl_stack_set_dev_flags)."""
        capi.check(self._lib.nl_stack_set_dev_flags(self._h, int(flags)))

    @property
    def last_fallback_pixels(self):
        return int(self._lib.nl_stack_last_fallback_pixels(self._h))

    @property
    def last_pass_protocol(self):
        """Bit 0: fused protocol, bit 1: generic pass + first replay in one launch, bit 2: chunked (diagnostics)."""
        return int(self._lib.nl_stack_last_pass_protocol(self._h))

    @property
    def last_generic_pixels(self):
        return int(self._lib.nl_stack_last_generic_pixels(self._h))

    @property
    def linfit_stage_counts(self):
        """list lengths of the last linear-fit cascade (see include/nlstack.h); [] if none ran"""
        buf = (C.c_uint * 8)()
        k = int(self._lib.nl_stack_linfit_stage_counts(self._h, buf, 8))
        return [int(buf[i]) for i in range(max(k, 0))]

    @property
    def last_mode(self):
        return self._lib.nl_stack_last_mode(self._h)

    @property
    def last_kernel_ms(self):
        return float(self._lib.nl_stack_last_kernel_ms(self._h))

    @property
    def last_dominant_kernel_ms(self):
        return float(self._lib.nl_stack_last_dominant_kernel_ms(self._h))

    @property
    def last_kernel_name(self):
        return self._lib.nl_stack_last_kernel_name(self._h).decode()

    def pass_times(self, back=0):
        """(pass ms, dominant-kernel ms) of the pass enqueued `back` passes ago, from the
        handle's ring of HIP events -- no host sync was needed while the passes were queued."""
        p, d = C.c_float(), C.c_float()
        capi.check(self._lib.nl_stack_pass_times(self._h, int(back), C.byref(p), C.byref(d)))
        return float(p.value), float(d.value)

    @property
    def stream_ptr(self):
        """hipStream_t of the handle as an integer (torch.cuda.ExternalStream takes it)."""
        return int(self._lib.nl_stack_stream(self._h) or 0)

    def copy_counters_async(self, device_ptr):
        """Enqueue a copy of the last pass's counters to a device buffer (2 x int64) on the handle's stream."""
        capi.check(self._lib.nl_stack_copy_counters_async(self._h, C.c_void_p(int(device_ptr))))

    def set_counters_buffer(self, device_ptr):
        """Passes enqueued from now on leave their counters in the caller's device buffer (32 bytes; None: the handle's own)."""
        capi.check(self._lib.nl_stack_set_counters_buffer(self._h, C.c_void_p(int(device_ptr)) if device_ptr else None))

    def order_stream_after(self, hip_stream):
        """`hip_stream` (integer hipStream_t) waits for everything enqueued on the handle so far."""
        capi.check(self._lib.nl_stack_order_stream_after(self._h, C.c_void_p(int(hip_stream))))

    @property
    def counters_device_ptr(self):
        """Device address of the last pass's {clip_low, clip_high} (2 x int64)."""
        return int(self._lib.nl_stack_counters_device_ptr(self._h) or 0)

    def find_sigmas(self, mode, clip_perc_low, clip_perc_high, ref_loc=0.0, reduce=None,
                    fetch=True):
        """Goal-seek bisection (stackfindsigma.go:48-98).  `reduce(lo, hi) ->
        (lo, hi)` maps this tile's counters to the totals over all tiles."""
        cl, ch = C.c_int64(0), C.c_int64(0)
        sl, sh, passes = C.c_float(), C.c_float(), C.c_int()
        out = np.zeros(self.width * self.height, np.float32) if fetch else None

        def _cb(counters, _user):
            try:
                lo, hi = reduce(int(counters[0]), int(counters[1]))
                counters[0], counters[1] = int(lo), int(hi)
                return 0
            except Exception:   # never unwind through the C frame
                return 1

        cb = capi.REDUCE_FN(_cb) if reduce is not None else C.cast(None, capi.REDUCE_FN)
        capi.check(self._lib.nl_stack_find_sigmas(
            self._h, int(mode), C.c_float(ref_loc), C.c_float(clip_perc_low),
            C.c_float(clip_perc_high), cb, None, capi.fptr(out) if fetch else None,
            C.byref(cl), C.byref(ch), C.byref(sl), C.byref(sh), C.byref(passes)))
        return out, cl.value, ch.value, float(sl.value), float(sh.value), passes.value

    # -- stack of stacks ---------------------------------------------------
    def accumulate(self, weight, first):
        capi.check(self._lib.nl_stack_accumulate(self._h, C.c_float(weight), int(bool(first))))

    def accumulate_finalize(self, weight_sum):
        out = np.zeros(self.width * self.height, np.float32)
        capi.check(self._lib.nl_stack_accumulate_finalize(self._h, C.c_float(weight_sum),
                                                          capi.fptr(out)))
        return out

    # -- per-frame statistics ----------------------------------------------
    def frame_stats(self, idx, variance=True):
        mn, mean, mx = C.c_float(), C.c_float(), C.c_float()
        var = C.c_double()
        capi.check(self._lib.nl_stack_frame_stats(self._h, int(idx), C.byref(mn), C.byref(mean),
                                                  C.byref(mx), C.byref(var) if variance else None))
        return (np.float32(mn.value), np.float32(mean.value), np.float32(mx.value),
                float(var.value) if variance else None)

    def frame_noise(self, idx):
        v = C.c_float()
        capi.check(self._lib.nl_stack_frame_noise(self._h, int(idx), C.byref(v)))
        return np.float32(v.value)

    def weights_from_noise(self):
        noise = np.zeros(self.n_frames, np.float32)
        capi.check(self._lib.nl_stack_weights_from_noise(self._h, capi.fptr(noise)))
        return noise


    # ---- formats and steps either side of the stack (include/nlstack.h, F3 / F4) ----
    def upload_frame_fits(self, idx, raw, bitpix, bscale=1.0, bzero=0.0, multiplier=1.0, offset=0.0):
        """Big-endian FITS payload bytes of this handle's tile -> frame slot idx,
        decoded on the device; returns (min, max, mean) of the decoded tile."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        stats = np.zeros(3, np.float32)
        capi.check(self._lib.nl_stack_upload_frame_fits(
            self._h, int(idx), raw.ctypes.data_as(C.c_void_p), int(bitpix), float(bscale), float(bzero),
            float(multiplier), float(offset), capi.fptr(stats)))
        return stats

    def upload_frame_projected(self, idx, src, src_w, src_h, trans, out_of_bounds=float("nan"),
                               multiplier=1.0, offset=0.0):
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1)
        t = np.ascontiguousarray(trans, dtype=np.float32).reshape(6)
        capi.check(self._lib.nl_stack_upload_frame_projected(
            self._h, int(idx), capi.fptr(src), int(src_w), int(src_h), capi.fptr(t), float(out_of_bounds),
            float(multiplier), float(offset)))

    def upload_frame_fits_async(self, idx, raw, bitpix, bscale=1.0, bzero=0.0, multiplier=1.0, offset=0.0):
        """Overlapped form of upload_frame_fits (pinned ring, copy stream, no statistics)."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        capi.check(self._lib.nl_stack_upload_frame_fits_async(
            self._h, int(idx), raw.ctypes.data_as(C.c_void_p), int(bitpix), float(bscale), float(bzero),
            float(multiplier), float(offset)))

    def upload_frame_projected_async(self, idx, src, src_w, src_h, trans, out_of_bounds=float("nan"),
                                     multiplier=1.0, offset=0.0):
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1)
        t = np.ascontiguousarray(trans, dtype=np.float32).reshape(6)
        capi.check(self._lib.nl_stack_upload_frame_projected_async(
            self._h, int(idx), capi.fptr(src), int(src_w), int(src_h), capi.fptr(t), float(out_of_bounds),
            float(multiplier), float(offset)))

    def frame_affine(self, idx, multiplier, offset):
        capi.check(self._lib.nl_stack_frame_affine(self._h, int(idx), float(multiplier), float(offset)))

    def download_result_fits(self):
        raw = np.empty(self.tile_pixels * 4, np.uint8)
        capi.check(self._lib.nl_stack_download_result_fits(self._h, raw.ctypes.data_as(C.c_void_p)))
        return raw


class StackGroup:
    """nl_group_*: one stack over several GPUs from one process -- tile t owns the rows
    group_tile_rows(height, n_tiles, t) of all frames on devices[t]; counters summed on the
    host (stack.go:142-152, 193-198).  devices=None: one tile per visible device."""

    def __init__(self, n_frames, width, height, n_tiles=0, devices=None):
        self._lib = capi.load()
        self.n_frames, self.width, self.height = int(n_frames), int(width), int(height)
        dev = None
        if devices is not None:
            n_tiles = len(devices)
            dev = (C.c_int * n_tiles)(*[int(d) for d in devices])
        self._g = self._lib.nl_group_create(self.n_frames, self.width, self.height, int(n_tiles), dev)
        if not self._g:
            raise capi.NlError(capi.ERR_HIP, capi.last_error())
        self.size = self._lib.nl_group_size(self._g)

    def close(self):
        if self._g:
            self._lib.nl_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def tile_rows(self, t):
        r0, nr = C.c_int(), C.c_int()
        self._lib.nl_group_tile_rows(self.height, self.size, int(t), C.byref(r0), C.byref(nr))
        return r0.value, nr.value

    def tile(self, t):
        """Borrowed view of tile t's handle (nl_group_tile): the group keeps ownership."""
        r0, nr = self.tile_rows(t)
        h = self._lib.nl_group_tile(self._g, int(t))
        if not h:
            raise IndexError(t)
        return StackHandle._borrow(h, self.n_frames, self.width, self.height, r0, nr)

    def upload_frames(self, frames):
        for i, f in enumerate(frames):
            f = np.ascontiguousarray(f, dtype=np.float32).reshape(-1)
            assert f.size == self.width * self.height
            capi.check(self._lib.nl_group_upload_frame(self._g, i, capi.fptr(f)))

    def upload_frame_fits(self, idx, raw, bitpix, bscale=1.0, bzero=0.0, multiplier=1.0, offset=0.0):
        """raw: big-endian FITS payload of the WHOLE frame; every tile decodes its rows on its device."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        capi.check(self._lib.nl_group_upload_frame_fits(
            self._g, int(idx), raw.ctypes.data_as(C.c_void_p), int(bitpix), float(bscale), float(bzero),
            float(multiplier), float(offset)))

    def upload_frame_projected(self, idx, src, src_w, src_h, trans, out_of_bounds=float("nan"),
                               multiplier=1.0, offset=0.0):
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1)
        t = np.ascontiguousarray(trans, dtype=np.float32).reshape(6)
        capi.check(self._lib.nl_group_upload_frame_projected(
            self._g, int(idx), capi.fptr(src), int(src_w), int(src_h), capi.fptr(t), float(out_of_bounds),
            float(multiplier), float(offset)))

    def fill_synthetic(self, seed=0x4E4C5354):
        capi.check(self._lib.nl_group_fill_synthetic(self._g, C.c_uint64(seed)))

    def set_weights(self, weights):
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        capi.check(self._lib.nl_group_set_weights(self._g, capi.fptr(w) if w is not None else None))

    def set_exact(self, on=True):
        capi.check(self._lib.nl_group_set_exact(self._g, int(on)))

    def run(self, mode, sigma_low=2.75, sigma_high=2.75, ref_loc=0.0, download=True):
        """download=False: the result stays on the devices (e.g. for accumulate); returns (None, low, high)."""
        out = np.zeros(self.width * self.height, np.float32) if download else None
        cl, ch = C.c_int64(0), C.c_int64(0)
        capi.check(self._lib.nl_group_run(self._g, int(mode), C.c_float(sigma_low), C.c_float(sigma_high),
                                          C.c_float(ref_loc), capi.fptr(out) if download else None,
                                          C.byref(cl), C.byref(ch)))
        return out, cl.value, ch.value

    def upload_frame(self, idx, frame):
        """Overlapped upload of one whole frame (nl_group_upload_frame)."""
        f = np.ascontiguousarray(frame, dtype=np.float32).reshape(-1)
        assert f.size == self.width * self.height
        capi.check(self._lib.nl_group_upload_frame(self._g, int(idx), capi.fptr(f)))

    def find_sigmas(self, mode, clip_perc_low, clip_perc_high, ref_loc=0.0):
        out = np.zeros(self.width * self.height, np.float32)
        cl, ch = C.c_int64(0), C.c_int64(0)
        sl, sh, passes = C.c_float(), C.c_float(), C.c_int()
        capi.check(self._lib.nl_group_find_sigmas(
            self._g, int(mode), C.c_float(ref_loc), C.c_float(clip_perc_low), C.c_float(clip_perc_high),
            capi.fptr(out), C.byref(cl), C.byref(ch), C.byref(sl), C.byref(sh), C.byref(passes)))
        return out, cl.value, ch.value, float(sl.value), float(sh.value), passes.value

    def accumulate(self, weight, first):
        capi.check(self._lib.nl_group_accumulate(self._g, C.c_float(weight), int(bool(first))))

    def accumulate_finalize(self, weight_sum):
        out = np.zeros(self.width * self.height, np.float32)
        capi.check(self._lib.nl_group_accumulate_finalize(self._g, C.c_float(weight_sum), capi.fptr(out)))
        return out


def fits_parse_header(file_bytes, frame_id=0):
    """Header of a FITS file image (read.go:445-469, 97-147): dict with bitpix, naxisn, bzero, bscale, exposure,
    pixels, header_bytes (= offset of the payload), payload_bytes, padded_payload_bytes."""
    lib = capi.load()
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    h = capi.FitsHeader()
    capi.check(lib.nl_fits_parse_header(buf.ctypes.data_as(C.c_void_p), buf.size, int(frame_id), C.byref(h)))
    return {"bitpix": h.bitpix, "naxisn": [h.naxisn[i] for i in range(h.naxis)], "bzero": np.float32(h.bzero),
            "bscale": np.float32(h.bscale), "exposure": np.float32(h.exposure), "pixels": h.pixels,
            "header_bytes": h.header_bytes, "payload_bytes": h.payload_bytes,
            "padded_payload_bytes": h.padded_payload_bytes}


def fits_write_header(naxisn, bzero=0.0, bscale=1.0, exposure=0.0):
    """The header Image.Write emits for a BITPIX -32 image (write.go:54-89), padded to 2880-byte blocks."""
    lib = capi.load()
    ax = (C.c_int32 * len(naxisn))(*[int(n) for n in naxisn])
    n = lib.nl_fits_write_header(None, 0, len(naxisn), ax, float(bzero), float(bscale), float(exposure))
    if n < 0:
        raise capi.NlError(capi.ERR_INVALID_ARG, capi.last_error())
    out = np.empty(n, np.uint8)
    got = lib.nl_fits_write_header(out.ctypes.data_as(C.c_void_p), n, len(naxisn), ax, float(bzero), float(bscale),
                                   float(exposure))
    assert got == n
    return out.tobytes()


def fits_padded_bytes(payload_bytes):
    return int(capi.load().nl_fits_padded_bytes(int(payload_bytes)))


def fits_decode(raw, bitpix, bscale=1.0, bzero=0.0, device=0):
    lib = capi.load()
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    bpv = {8: 1, 16: 2, 32: 4, 64: 8, -32: 4, -64: 8}.get(int(bitpix), 1)
    n = raw.size // bpv
    out = np.empty(n, np.float32)
    stats = np.zeros(3, np.float32)
    capi.check(lib.nl_fits_decode(raw.ctypes.data_as(C.c_void_p), int(bitpix), n, float(bscale),
                                  float(bzero), capi.fptr(out), capi.fptr(stats), int(device)))
    return out, stats


def project_bilinear(src, src_w, src_h, dst_w, dst_h, trans, out_of_bounds=float("nan"), device=0):
    lib = capi.load()
    src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1)
    t = np.ascontiguousarray(trans, dtype=np.float32).reshape(6)
    dst = np.empty(int(dst_w) * int(dst_h), np.float32)
    capi.check(lib.nl_project_bilinear(capi.fptr(src), int(src_w), int(src_h), capi.fptr(dst), int(dst_w),
                                       int(dst_h), capi.fptr(t), float(out_of_bounds), int(device)))
    return dst


def weights_from_scalars(weighting, per_frame):
    """getWeights (stack.go:231-270) on per-frame exposure / noise / HFR."""
    lib = capi.load()
    pf = np.ascontiguousarray(per_frame, dtype=np.float32)
    w = np.zeros(pf.size, np.float32)
    bad = C.c_int(-1)
    rc = lib.nl_weights_from_scalars(int(weighting), capi.fptr(pf), pf.size, capi.fptr(w),
                                     C.byref(bad))
    if rc != capi.OK:
        raise capi.NlError(rc, capi.last_error())
    return None if weighting == capi.WEIGHT_NONE else w


def median_filter_3x3(image, width, height, device=0):
    lib = capi.load()
    src = np.ascontiguousarray(image, dtype=np.float32).reshape(-1)
    dst = np.empty_like(src)
    capi.check(lib.nl_median_filter_3x3(capi.fptr(src), capi.fptr(dst), int(width), int(height),
                                        int(device)))
    return dst


def median_filter_mask(data, mask, device=0):
    """MedianFilter (ops/pre/badpixels.go:54-77): out[i] = median of data[i + mask[j]] inside the data."""
    lib = capi.load()
    src = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
    m = np.ascontiguousarray(mask, dtype=np.int32)
    dst = np.empty_like(src)
    capi.check(lib.nl_median_filter_mask(capi.fptr(src), capi.fptr(dst), src.size,
                                         m.ctypes.data_as(C.POINTER(C.c_int32)), m.size, int(device)))
    return dst
