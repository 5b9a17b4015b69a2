"""GPU parity of the per-frame kernels and the stack-of-stacks accumulator
through the C ABI -- the HIP replacements of the reference's only native code:

  nl_stack_frame_stats         calcMinMeanMaxAVX2 / calcVarianceAVX2  internal/stats/stats_amd64.s:28-143,
                               pure Go twins internal/stats/stats.go:264-287
  nl_stack_frame_noise         estimateNoisePureGo  internal/stats/noise.go:32-55
                               (AVX2 twin: noise_amd64.s:78-195)
  nl_stack_weights_from_noise  getWeights, inverse noise  internal/ops/stack/stack.go:241-253
  nl_median_filter_3x3         MedianFilter3x3  internal/median/median3x3.go:26-110
  nl_stack_accumulate(_finalize)  StackIncremental(+Finalize)  internal/ops/stack/stack.go:924-944

Bars.  min / max, the 3x3 median and the accumulator are order independent or
elementwise: bit-exact.  The frame sums are fp64 (mean, variance) or fp32
(noise) accumulations whose ORDER already differs between the reference's own
two code paths (pure Go: sequential; AVX2: 4 resp. 8 lanes), so a tolerance
is stated per quantity below and the device value must sit within it of BOTH
oracle orders; on inputs whose sums are exact in every order (small integers)
the device value must be bit-identical.
"""
import numpy as np
import pytest

from util import bits_equal, make_frames

pytestmark = pytest.mark.gpu

# fp64 accumulation of <= 2^24 fp32 values: any two summation orders agree to
# n*2^-53 relative (all terms of the variance sum are >= 0; the mean's terms here
# are dominated by a positive background) -- 1e-9 leaves three orders of margin.
F64_ORDER_RTOL = 1e-9
# the mean is that fp64 sum rounded once to fp32: two orders can land on
# neighbouring fp32 values only when the fp64 value sits on a rounding boundary
MEAN_ULPS = 1
# noise: the reference sums |conv| in fp32, sequentially per row then over rows
# (noise.go:40-50); its AVX2 twin in 8 lanes with FMA.  The device sums in fp64
# (closer to the true sum than either).  Sequential fp32 summation of n positive
# terms is off by at most n*2^-24 relative per level (rows of <= 4094, <= 4094
# rows), typically sqrt(n)*2^-24 ~ 4e-6: bar 5e-5.
NOISE_RTOL = 5e-5


def natural_image(width, height, seed, nan_free=True):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    img = 1000.0 + 150.0 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 30.0 * rng.standard_normal((height, width))
    hot = rng.random((height, width)) < 0.002
    img[hot] += 5000.0 * rng.random(np.count_nonzero(hot))
    return img.astype(np.float32).reshape(-1)


def ulp_distance(a, b):
    ia = np.array([a], np.float32).view(np.int32)[0]
    ib = np.array([b], np.float32).view(np.int32)[0]
    return abs(int(ia) - int(ib))


# ---- A10b: min / mean / max / variance ----------------------------------------

@pytest.mark.parametrize("width,height", [(64, 8), (67, 29), (1024, 1024), (4096, 1024), (4100, 515)])
def test_frame_stats_matches_both_reference_orders(nl, oracle, width, height):
    frames = np.stack([natural_image(width, height, 11 + k) for k in range(3)])
    frames[1] -= np.float32(2000.0)          # a frame with negative values
    with nl.StackHandle(3, width, height) as st:
        st.upload_frames(frames)
        for k in range(3):
            mn, mean, mx, var = st.frame_stats(k)
            # min / max: order independent, bit-exact
            omn, omean, omx = oracle.min_mean_max(frames[k])
            assert bits_equal(np.array([mn, mx]), np.array([omn, omx])), (k, mn, omn, mx, omx)
            assert bits_equal(np.array([mn, mx]), np.array([frames[k].min(), frames[k].max()]))
            refs = [(omean, oracle.variance(frames[k], mean))]
            if (width * height) % 4 == 0:    # the AVX2 routine needs a multiple of 4 (stats.go:266)
                lmn, lmean, lmx = oracle.min_mean_max(frames[k], lanes4=True)
                assert bits_equal(np.array([mn, mx]), np.array([lmn, lmx]))
                refs.append((lmean, oracle.variance(frames[k], mean, lanes4=True)))
            for rmean, rvar in refs:
                assert ulp_distance(mean, rmean) <= MEAN_ULPS, (k, mean, rmean)
                assert abs(var - rvar) <= F64_ORDER_RTOL * rvar, (k, var, rvar)


def test_frame_stats_exact_when_sums_are_exact(nl, oracle):
    # small integers: every partial sum is an integer < 2^53, so all orders agree and
    # the device must be bit-identical in the mean and the variance
    width, height = 512, 384
    rng = np.random.default_rng(3)
    frame = rng.integers(-500, 4000, width * height).astype(np.float32)
    with nl.StackHandle(1, width, height) as st:
        st.upload_frames(frame[None])
        mn, mean, mx, var = st.frame_stats(0)
    omn, omean, omx = oracle.min_mean_max(frame)
    assert (mn, mx) == (omn, omx)
    assert bits_equal(np.array([mean]), np.array([omean]))
    # variance terms (x - mean)^2 are not integers; the sum of integers check is the mean.
    assert abs(var - oracle.variance(frame, mean)) <= F64_ORDER_RTOL * var


def test_frame_stats_nan_semantics(nl, oracle):
    # a comparison with NaN is false in the reference (stats.go:268-271): NaN never becomes
    # min / max, but poisons the mean
    width, height = 96, 40
    frame = natural_image(width, height, 5)
    frame[1234] = np.nan
    with nl.StackHandle(1, width, height) as st:
        st.upload_frames(frame[None])
        mn, mean, mx, _ = st.frame_stats(0)
    omn, omean, omx = oracle.min_mean_max(frame)
    assert (mn, mx) == (omn, omx) and np.isnan(mean) and np.isnan(omean)


def test_frame_stats_on_a_row_tile(nl, oracle):
    # per-tile statistics of a sharded stack: the tile's own pixels only
    width, height, row0, rows = 300, 64, 24, 17
    frame = natural_image(width, height, 8)
    with nl.StackHandle(1, width, height, row0=row0, rows=rows) as st:
        st.upload_frame(0, frame)
        mn, mean, mx, var = st.frame_stats(0)
    tile = frame[row0 * width:(row0 + rows) * width]
    omn, omean, omx = oracle.min_mean_max(tile)
    assert (mn, mx) == (omn, omx) and ulp_distance(mean, omean) <= MEAN_ULPS
    assert abs(var - oracle.variance(tile, mean)) <= F64_ORDER_RTOL * var


# ---- A10: EstimateNoise --------------------------------------------------------

@pytest.mark.parametrize("width,height", [(3, 3), (4, 3), (3, 9), (64, 8), (67, 29), (1024, 1024), (4096, 1024)])
def test_frame_noise_matches_oracle(nl, oracle, width, height):
    frames = np.stack([natural_image(width, height, 21 + k) for k in range(2)])
    with nl.StackHandle(2, width, height) as st:
        st.upload_frames(frames)
        for k in range(2):
            got = st.frame_noise(k)
            want = oracle.estimate_noise(frames[k], width)
            assert abs(float(got) - float(want)) <= NOISE_RTOL * abs(float(want)), (k, got, want)


def test_frame_noise_per_pixel_convolution_is_bit_identical(nl, oracle):
    # integer pixels < 2^10: every product and partial sum of the convolution and every
    # partial sum of |conv| over the image is an integer < 2^24, i.e. exact in fp32 in any
    # order -- so device (fp64 sum) and oracle (fp32 sequential sums) must agree bit for bit.
    # A wrong tap, weight or sign in the kernel's stencil changes the integer total.
    width, height = 200, 120
    rng = np.random.default_rng(77)
    frame = rng.integers(0, 37, width * height).astype(np.float32)     # sum |conv| < 198*118*16*36 < 2^24
    with nl.StackHandle(1, width, height) as st:
        st.upload_frames(frame[None])
        got = st.frame_noise(0)
    want = oracle.estimate_noise(frame, width)
    assert bits_equal(np.array([got]), np.array([want])), (got, want)
    # asymmetric content: a transposed or mirrored stencil would still pass on symmetric noise,
    # a ramp + single hot pixel pins row / column orientation of the +-1/-2/4 taps through the borders
    frame = (np.arange(width * height) % 23).astype(np.float32)
    frame[5 * width + 1] = 900.0            # hot pixel in the first interior column
    frame[1 * width + 7] = 700.0            # ... and in the first interior row
    with nl.StackHandle(1, width, height) as st:
        st.upload_frames(frame[None])
        got = st.frame_noise(0)
    assert bits_equal(np.array([got]), np.array([oracle.estimate_noise(frame, width)]))


def test_frame_noise_rejects_row_tiles_and_tiny_images(nl):
    from nightlight_amd import capi
    with nl.StackHandle(1, 64, 64, row0=8, rows=16) as st:
        with pytest.raises(capi.NlError):
            st.frame_noise(0)
    with nl.StackHandle(1, 2, 8) as st:
        with pytest.raises(capi.NlError):
            st.frame_noise(0)


# ---- F1: noise -> inverse-noise weights -> weighted stack, on the device -----------

@pytest.mark.parametrize("mode", [1, 2, 3])
def test_weights_from_noise_then_weighted_stack(nl, oracle, mode):
    n, width, height = 24, 160, 48
    frames = make_frames(n, width, height, seed=640, nan_frac=0.0, nan_border=False, all_nan_patch=False)
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        noise = st.weights_from_noise()
        got, cl, ch = st.run(mode, 2.75, 2.75)
    # (a) the device's per-frame noise against the oracle's EstimateNoise
    want_noise = np.array([oracle.estimate_noise(frames[k], width) for k in range(n)], np.float32)
    assert np.all(np.abs(noise.astype(np.float64) - want_noise) <= NOISE_RTOL * np.abs(want_noise))
    assert np.unique(noise).size > 3          # make_frames gives frames distinct noise levels
    # (b) getWeights (stack.go:246-253) on those noise values, then the weighted stack: bit-exact
    rc, w, _ = oracle.get_weights(oracle.WEIGHT_INVERSE_NOISE, noise)
    assert rc == 0
    rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, w, 2.75, 2.75, 0.0, num_cpu=4)
    assert rc == 0
    assert bits_equal(got, want)
    if mode >= 2:
        assert (cl, ch) == (wl, wh)
    # (c) end to end with the oracle's own noise values: weights move by <= a few 1e-5
    # relative, the weighted mean by less; clipping does not depend on the weights at all
    rc, w2, _ = oracle.get_weights(oracle.WEIGHT_INVERSE_NOISE, want_noise)
    rc, want2, wl2, wh2, _ = oracle.stack_apply(mode, frames, w2, 2.75, 2.75, 0.0, num_cpu=4)
    assert (wl2, wh2) == (wl, wh)
    assert np.max(np.abs(got.astype(np.float64) - want2) / np.abs(want2)) <= 1e-5


def test_weights_from_noise_with_nan_borders(nl, oracle):
    # aligned frames carry NaN borders; the reference computes noise BEFORE alignment on
    # NaN-free data, but getWeights must still behave: NaN noise -> NaN weights, as the oracle
    n, width, height = 6, 64, 32
    frames = make_frames(n, width, height, seed=641, nan_frac=0.0, nan_border=False, all_nan_patch=False)
    frames[2, 700] = np.nan
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        noise = st.weights_from_noise()
    assert np.isnan(noise[2]) and np.isnan(oracle.estimate_noise(frames[2], width))
    assert not np.isnan(np.delete(noise, 2)).any()


# ---- A15: 3x3 median filter ------------------------------------------------------

@pytest.mark.parametrize("width,height", [(1, 1), (1, 7), (7, 1), (2, 2), (3, 3), (3, 40), (40, 3), (4, 4),
                                          (67, 29), (256, 64), (1031, 517), (4096, 512)])
def test_median_filter_3x3_bit_exact(nl, oracle, width, height):
    img = natural_image(width, height, 31)
    got = nl.median_filter_3x3(img, width, height)
    want = oracle.median_filter_3x3(img, width)
    assert bits_equal(got, want), "%dx%d: %d pixels differ" % (
        width, height, np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
    if width >= 3 and height >= 3:            # independent check of the interior: numpy's median of the 9
        g = img.reshape(height, width)
        nine = np.stack([g[dy:height - 2 + dy, dx:width - 2 + dx] for dy in range(3) for dx in range(3)])
        assert np.array_equal(got.reshape(height, width)[1:-1, 1:-1], np.median(nine, axis=0))


def test_median_filter_3x3_ties_and_signed_zero(nl, oracle):
    width, height = 128, 48
    rng = np.random.default_rng(9)
    img = rng.integers(-2, 3, width * height).astype(np.float32)      # heavy ties, 0 and values around it
    got = nl.median_filter_3x3(img, width, height)
    want = oracle.median_filter_3x3(img, width)
    assert np.array_equal(got, want)


# ---- A14: StackIncremental / StackIncrementalFinalize -------------------------------

def test_accumulate_matches_stack_incremental(nl, oracle):
    # three batches of different sizes, each stacked on the device, combined on the device with
    # weights = batch frame counts (stackbatches.go:97-116), against the oracle's loop: bit-exact
    width, height = 97, 41
    sizes = [7, 5, 9]
    batches = [make_frames(s, width, height, seed=800 + i) for i, s in enumerate(sizes)]
    acc_want = np.zeros(width * height, np.float32)
    acc = None
    for i, (s, frames) in enumerate(zip(sizes, batches)):
        rc, res, _, _, _ = oracle.stack_apply(2, frames, None, 2.75, 2.75, 0.0, num_cpu=2)
        assert rc == 0
        acc_want = oracle.stack_incremental(acc_want, res, float(s), first=(i == 0))
    want = oracle.stack_incremental_finalize(acc_want, float(sum(sizes)))

    # one handle sized for the largest batch holds the accumulator across batches
    with nl.StackHandle(max(sizes), width, height) as st:
        for i, (s, frames) in enumerate(zip(sizes, batches)):
            pad = np.full((max(sizes), width * height), np.nan, np.float32)   # missing frames = NaN = no data
            pad[:s] = frames
            st.upload_frames(pad)
            st.set_exact(True)
            st.run(2, 2.75, 2.75, fetch=False)
            st.accumulate(float(s), first=(i == 0))
        got = st.accumulate_finalize(float(sum(sizes)))
    assert bits_equal(got, want), "%d pixels differ" % np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))


def test_accumulate_on_a_row_tile_and_error_before_first(nl, oracle):
    from nightlight_amd import capi
    width, height, row0, rows = 64, 48, 16, 20
    frames = make_frames(4, width, height, seed=801)
    with nl.StackHandle(4, width, height, row0=row0, rows=rows) as st:
        with pytest.raises(capi.NlError):
            st.accumulate_finalize(1.0)
        st.upload_frames(frames)
        st.run(1, fetch=False)
        st.accumulate(3.0, first=True)
        st.run(1, fetch=False)
        st.accumulate(5.0, first=False)
        got = st.accumulate_finalize(8.0)
    rc, res, _, _, _ = oracle.stack_apply(1, frames, None)
    acc = oracle.stack_incremental(np.zeros_like(res), res, 3.0, first=True)
    acc = oracle.stack_incremental(acc, res, 5.0, first=False)
    want = oracle.stack_incremental_finalize(acc, 8.0)
    sl = slice(row0 * width, (row0 + rows) * width)
    assert bits_equal(got[sl], want[sl])
    assert not got[:row0 * width].any() and not got[(row0 + rows) * width:].any()   # other rows untouched


# ---- download_rows (used by the full-size tests) ---------------------------------

def test_download_rows(nl):
    from nightlight_amd import capi
    width, height, row0, rows, n = 50, 40, 8, 24, 3
    frames = make_frames(n, width, height, seed=802)
    with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
        st.upload_frames(frames)
        got = st.download_rows(2, 5, 7)
        assert bits_equal(got, frames[2][(row0 + 5) * width:(row0 + 12) * width])
        res, _, _ = st.run(1)
        assert bits_equal(st.download_rows(-1, 0, rows), res[row0 * width:(row0 + rows) * width])
        for bad in ((n, 0, 1), (-2, 0, 1), (0, rows, 1), (0, -1, 1), (0, 0, rows + 1), (0, 3, 0)):
            with pytest.raises(capi.NlError):
                st.download_rows(*bad)


# ---- A16: MedianFloat32 / GatherAndMedian / MedianFilter ---------------------------------

@pytest.mark.parametrize("radius", [1.0, 1.5, 2.0, 2.5])
@pytest.mark.parametrize("width,height", [(64, 16), (211, 37)])
def test_median_filter_mask_matches_gather_and_median(nl, oracle, radius, width, height):
    # star.CreateMask discs of 5 .. 21 offsets (findstars.go:187-200; 9 offsets at radius 1.5 take the
    # reference's 9-value network).  Where the whole neighbourhood lies inside the data the reference's
    # GatherAndMedian is history-free and the device must match it bit for bit; elsewhere the device
    # returns the median of the values that exist (gather.go:37 reads stale buffer contents there).
    img = natural_image(width, height, 55)
    img[::7] = np.round(img[::7] / 64.0) * 64.0           # ties
    mask = oracle.create_mask(width, radius)
    assert mask.size in (5, 9, 13, 21)
    got = nl.median_filter_mask(img, mask)
    want, full = oracle.median_filter_mask(img, mask)
    assert full.sum() == img.size - (int(mask.max()) - int(mask.min()))     # linear-index neighbourhoods
    assert bits_equal(got[full], want[full]), "%d pixels differ" % np.count_nonzero(got[full] != want[full])
    # edges: median of the existing neighbours, even counts averaged (qsort.go:68-82)
    for i in np.flatnonzero(~full)[:: max(1, (~full).sum() // 200)]:
        idx = i + mask
        vals = np.sort(img[idx[(idx >= 0) & (idx < img.size)]])
        k = vals.size // 2
        ref = vals[k] if vals.size % 2 else np.float32(0.5) * (vals[k - 1] + vals[k])
        assert got[i] == ref, (i, got[i], ref)


def test_median_filter_mask_argument_errors(nl):
    from nightlight_amd import capi
    img = np.zeros(64, np.float32)
    with pytest.raises(capi.NlError):
        nl.median_filter_mask(img, np.arange(40, dtype=np.int32))          # more than 32 offsets
    with pytest.raises(capi.NlError):
        nl.median_filter_mask(img, np.zeros(0, np.int32))
