"""GPU parity for the formats / steps either side of the stack (include/nlstack.h,
rows F3 / F4 of SURVEY section 8f), through the C ABI, against the CPU oracle:
FITS payload decode (fits/read.go:172-445), encode (fits/write.go:182-200),
MatchHistogram (fits/pixelops.go:601-605), Project (fits/project.go:26-76).
Everything is elementwise fp32 in the reference's operation order => bit-exact."""
import numpy as np
import pytest

from util import make_frames, same_values

pytestmark = pytest.mark.gpu

DT = {8: np.uint8, 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


def payload(bitpix, n, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(DT[bitpix])
    if bitpix > 0:
        info = np.iinfo(dt.newbyteorder("="))
        vals = rng.integers(info.min, info.max, n, dtype=dt.newbyteorder("="), endpoint=True)
    else:
        vals = (rng.standard_normal(n) * 1e3).astype(dt.newbyteorder("="))
        vals[::97] = np.nan
    return np.frombuffer(vals.astype(dt).tobytes(), np.uint8)


@pytest.mark.parametrize("bitpix", [8, 16, 32, 64, -32, -64])
@pytest.mark.parametrize("n", [1, 3, 4, 1021, 40000])
def test_fits_decode_is_bit_exact(nl, oracle, bitpix, n):
    from nightlight_amd import stack
    raw = payload(bitpix, n, seed=n + abs(bitpix))
    got, stats = stack.fits_decode(raw, bitpix, 0.5, 32768.0)
    rc, want, mn, mx, mean = oracle.fits_decode(raw, bitpix, 0.5, 32768.0)
    assert rc == 0 and same_values(got, want)
    assert stats[0] == np.float32(mn) and stats[1] == np.float32(mx)
    if np.isnan(mean):
        assert np.isnan(stats[2])
    else:                                     # fp64 sum in a different order, then one rounding to fp32
        assert abs(float(stats[2]) - mean) <= 1.2e-7 * abs(mean)


def test_unknown_bitpix_is_an_error(nl):
    from nightlight_amd import capi, stack
    with pytest.raises(capi.NlError) as e:
        stack.fits_decode(np.zeros(12, np.uint8), 24)
    assert "Unknown BITPIX value 24" in str(e.value)          # read.go:169


def test_stack_from_fits_payloads_with_fused_histogram_match(nl, oracle):
    # int16 payloads of a row tile -> frame slots (decode + x*m+o on the device) -> sigma clip;
    # result -> FITS payload bytes (big-endian, NaN -> 0)
    width, height, n = 96, 40, 12
    row0, rows = 8, 24
    rng = np.random.default_rng(5)
    raws = [payload(16, width * height, seed=50 + k) for k in range(n)]
    ms = rng.uniform(0.9, 1.1, n).astype(np.float32)
    os_ = rng.uniform(-20, 20, n).astype(np.float32)
    frames = []
    with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
        for k in range(n):
            tile = raws[k][row0 * width * 2:(row0 + rows) * width * 2]
            stats = st.upload_frame_fits(k, tile, 16, 1.0, 32768.0, ms[k], os_[k])
            rc, dec, mn, mx, mean = oracle.fits_decode(tile, 16, 1.0, 32768.0)
            assert (stats[0], stats[1]) == (np.float32(mn), np.float32(mx))
            assert abs(float(stats[2]) - mean) <= 1.2e-7 * abs(mean)
            want = oracle.affine(dec, ms[k], os_[k])
            assert same_values(st.download_tile(k), want)
            frames.append(want)
        st.set_exact(True)
        got, cl, ch = st.run(2, 2.0, 2.0)
        got = got[row0 * width:(row0 + rows) * width]          # run() returns the whole image, tile in place
        raw_out = st.download_result_fits()
    rc, want, wl, wh, _ = oracle.stack_apply(2, np.stack(frames), None, 2.0, 2.0)
    assert same_values(got, want) and (cl, ch) == (wl, wh)
    assert raw_out.tobytes() == oracle.fits_encode(want, True).tobytes()


def test_frame_affine_on_a_resident_frame(nl, oracle):
    width, height = 67, 13
    frames = make_frames(3, width, height, seed=8)
    with nl.StackHandle(3, width, height) as st:
        st.upload_frames(frames)
        st.frame_affine(1, 1.25, -3.5)
        assert same_values(st.download_tile(1), oracle.affine(frames[1], 1.25, -3.5))
        assert same_values(st.download_tile(0), frames[0].reshape(-1))


@pytest.mark.parametrize("trans", [[1, 0, 0.5, 0, 1, 0.25],
                                   [0.999, 0.03, -3.2, -0.03, 0.999, 4.7],
                                   [1.02, 0, 0, 0, 0.98, 0],
                                   [0, -1, 60, 1, 0, 0],
                                   [1, 0, 1e6, 0, 1, 0],            # everything out of bounds
                                   [1, 0, 0, 0, 1, 0]])
def test_project_is_bit_exact(nl, oracle, trans):
    from nightlight_amd import stack
    rng = np.random.default_rng(4)
    sw, sh, dw, dh = 131, 77, 140, 70
    src = rng.standard_normal(sw * sh).astype(np.float32)
    got = stack.project_bilinear(src, sw, sh, dw, dh, trans)
    rc, want = oracle.project_bilinear(src, sw, sh, dw, dh, trans, np.nan)
    assert rc == 0 and same_values(got, want)


def test_singular_transform_is_an_error(nl):
    from nightlight_amd import capi, stack
    with pytest.raises(capi.NlError) as e:
        stack.project_bilinear(np.zeros(16, np.float32), 4, 4, 4, 4, [1, 2, 0, 2, 4, 0])
    assert "Matrix has no inverse" in str(e.value)            # coord.go:160-163


def test_align_and_stack_tiles(nl, oracle):
    # each frame is projected with its own transform straight into its slot of two row-tile
    # handles (NaN where the aligned frame has no data), histogram-matched, then stacked
    sw = sh = 80
    width, height, n = 72, 64, 9
    rng = np.random.default_rng(6)
    srcs = [(1000 + 30 * rng.standard_normal(sw * sh)).astype(np.float32) for _ in range(n)]
    transs = [[1, 0.01 * (k - 4), 2.0 * k - 6.5, -0.01 * (k - 4), 1, 1.5 * k - 5.25] for k in range(n)]
    ms = rng.uniform(0.95, 1.05, n).astype(np.float32)
    os_ = rng.uniform(-5, 5, n).astype(np.float32)
    aligned = []
    for k in range(n):
        rc, a = oracle.project_bilinear(srcs[k], sw, sh, width, height, transs[k], np.nan)
        assert rc == 0
        aligned.append(oracle.affine(a, ms[k], os_[k]))
    rc, want, wl, wh, _ = oracle.stack_apply(2, np.stack(aligned), None, 2.5, 2.5)
    out = np.zeros(width * height, np.float32)
    tl = th = 0
    for row0, rows in ((0, 40), (40, 24)):
        with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
            for k in range(n):
                st.upload_frame_projected(k, srcs[k], sw, sh, transs[k], np.nan, ms[k], os_[k])
                assert same_values(st.download_tile(k), aligned[k][row0 * width:(row0 + rows) * width])
            st.set_exact(True)
            _, cl, ch = st.run(2, 2.5, 2.5, out=out)
            tl += cl
            th += ch
    assert same_values(out, want) and (tl, th) == (wl, wh)
    assert np.isnan(np.stack(aligned)).any()


def test_async_uploads_overlap_and_match_the_blocking_path(nl, oracle):
    # pinned staging ring (4 slots) + copy stream: more frames than slots, buffers reused and
    # overwritten on the host right after each call (the pointer must not be retained)
    width, height, n = 160, 96, 11
    frames = make_frames(n, width, height, seed=31)
    out = np.zeros(width * height, np.float32)
    tl = th = 0
    for row0, rows in ((0, 50), (50, 46)):
        with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
            scratch = np.empty(width * height, np.float32)
            for k in range(n):
                scratch[:] = frames[k].reshape(-1)
                st.upload_frame_async(k, scratch)
                scratch[:] = -1.0                      # caller reuses its buffer immediately
            st.set_exact(True)
            _, cl, ch = st.run(2, 2.5, 2.5, out=out)   # waits for the DMAs on the device
            tl += cl
            th += ch
            assert same_values(st.download_tile(n - 1), frames[n - 1].reshape(-1)[row0 * width:(row0 + rows) * width])
    rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, 2.5, 2.5)
    assert same_values(out, want) and (tl, th) == (wl, wh)


def test_group_ingest_fits_and_projected_through_the_pinned_ring(nl, oracle):
    # rows F3 / F4 on the multi-GPU drop-in (nl_group_upload_frame_fits / _projected): whole-frame payloads /
    # source frames in, every tile decodes / projects its own rows on its device, overlapped (more frames than
    # staging slots, the host buffers are overwritten right after each call); 3 tiles on this one device
    width, height, n = 88, 50, 10
    sw = sh = 96
    rng = np.random.default_rng(8)
    ms = rng.uniform(0.9, 1.1, n).astype(np.float32)
    os_ = rng.uniform(-20, 20, n).astype(np.float32)
    raws = [payload(16, width * height, seed=80 + k) for k in range(n)]
    decoded = []
    for k in range(n):
        rc, dec, _, _, _ = oracle.fits_decode(raws[k], 16, 1.0, 32768.0)
        decoded.append(oracle.affine(dec, ms[k], os_[k]))
    rc, want, wl, wh, _ = oracle.stack_apply(2, np.stack(decoded), None, 2.0, 2.0)
    with nl.StackGroup(n, width, height, n_tiles=3, devices=[0, 0, 0]) as g:
        scratch = np.empty(width * height * 2, np.uint8)
        for k in range(n):
            scratch[:] = raws[k]
            g.upload_frame_fits(k, scratch, 16, 1.0, 32768.0, ms[k], os_[k])
            scratch[:] = 0xAB
        g.set_exact(True)
        got, cl, ch = g.run(2, 2.0, 2.0)
        for t in range(3):
            r0, nr = g.tile_rows(t)
            assert same_values(g.tile(t).download_tile(n - 1), decoded[n - 1][r0 * width:(r0 + nr) * width])
    assert same_values(got, want) and (cl, ch) == (wl, wh)

    srcs = [(1000 + 30 * rng.standard_normal(sw * sh)).astype(np.float32) for _ in range(n)]
    transs = [[1, 0.012 * (k - 4), 2.0 * k - 9.5, -0.012 * (k - 4), 1, 1.5 * k - 7.25] for k in range(n)]
    aligned = []
    for k in range(n):
        rc, a = oracle.project_bilinear(srcs[k], sw, sh, width, height, transs[k], np.nan)
        assert rc == 0
        aligned.append(oracle.affine(a, ms[k], os_[k]))
    rc, want, wl, wh, _ = oracle.stack_apply(2, np.stack(aligned), None, 2.5, 2.5)
    with nl.StackGroup(n, width, height, n_tiles=3, devices=[0, 0, 0]) as g:
        scratch = np.empty(sw * sh, np.float32)
        for k in range(n):
            scratch[:] = srcs[k]
            g.upload_frame_projected(k, scratch, sw, sh, transs[k], np.nan, ms[k], os_[k])
            scratch[:] = -7.0
        g.set_exact(True)
        got, cl, ch = g.run(2, 2.5, 2.5)
    assert same_values(got, want) and (cl, ch) == (wl, wh)
    assert np.isnan(np.stack(aligned)).any()


def test_overlapped_ingest_on_one_handle_mixes_with_plain_uploads(nl, oracle):
    # nl_stack_upload_frame_fits_async / _projected_async / _async on the same ring, in any order
    width, height, n = 64, 30, 9
    row0, rows = 6, 20
    rng = np.random.default_rng(9)
    frames = []
    with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
        for k in range(n):
            if k % 3 == 0:
                raw = payload(-32, width * height, seed=90 + k)
                tile = raw[row0 * width * 4:(row0 + rows) * width * 4]
                st.upload_frame_fits_async(k, tile, -32, 1.0, 0.0)
                rc, dec, _, _, _ = oracle.fits_decode(raw, -32, 1.0, 0.0)
                frames.append(dec)
            elif k % 3 == 1:
                src = (500 + 10 * rng.standard_normal(70 * 40)).astype(np.float32)
                trans = [1, 0.02, 1.5 * k - 4, -0.02, 1, 0.5 * k - 3]
                st.upload_frame_projected_async(k, src, 70, 40, trans, np.nan, 1.0, 0.0)
                rc, a = oracle.project_bilinear(src, 70, 40, width, height, trans, np.nan)
                frames.append(a)
            else:
                f = (200 + 5 * rng.standard_normal(width * height)).astype(np.float32)
                st.upload_frame_async(k, f)
                frames.append(f)
        st.set_exact(True)
        got, cl, ch = st.run(1, 0.0, 0.0)              # mean: NaN = no data
        got = got[row0 * width:(row0 + rows) * width]
    tiles = np.stack(frames).reshape(n, height, width)[:, row0:row0 + rows, :].reshape(n, -1)
    rc, want, _, _, _ = oracle.stack_apply(1, np.ascontiguousarray(tiles), None, 0.0, 0.0)
    assert same_values(got, want)
