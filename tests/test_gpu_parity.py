"""GPU parity: every stacking mode through the C ABI (libnlstack.so) against
the CPU oracle on the same seeded inputs.  Bar: result bit-identical as fp32
values, clip counters equal (integers).  Mirrors the mode dispatch of
OpStack.Apply, internal/ops/stack/stack.go:156-190."""
import numpy as np
import pytest

from util import bits_equal, describe_mismatch, make_frames, same_values

pytestmark = pytest.mark.gpu

MODES = {0: "median", 1: "mean", 2: "sigma", 3: "winsor", 4: "mad", 5: "linearfit"}


# fp32 tolerance of the north star ("within 1e-5 relative"); only the
# register-resident sigma kernel needs it (its sums run in sorted order, the
# reference's in quickselect order) -- everything else is bit-exact.
RTOL = 1e-5


def close_values(a, b, rtol=RTOL):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    ok = ~np.isnan(a) & (a != b)          # equal values (incl. +-Inf) are fine as they are
    return bool(np.all(np.abs(a[ok].astype(np.float64) - b[ok]) <= rtol * np.abs(b[ok].astype(np.float64))))


def run_both(nl, oracle, mode, frames, width, height, weights, sl, sh, ref_loc=0.0, exact=True):
    """exact=True forces the bit-exact kernels; exact=False is the default
    dispatch (register-resident kernel for unweighted sigma clipping)."""
    n = frames.shape[0]
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        st.set_weights(weights)
        st.set_exact(exact)
        got, cl, ch = st.run(mode, sl, sh, ref_loc)
    ow = None if mode in (0, 5) else weights
    rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, ow, sl, sh, ref_loc, num_cpu=4)
    assert rc == 0
    return got, (cl, ch), want, (wl, wh)


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 15, 25, 64, 128])
def test_mode_matches_oracle(nl, oracle, mode, n):
    width, height = 67, 29            # 1943 pixels: ragged vs 64-lane tiles and vs float4
    frames = make_frames(n, width, height, seed=100 + n)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, None, 2.75, 2.75)
    assert same_values(got, want), "%s n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
    if mode >= 2:
        assert gc == wc, "%s n=%d clip counters %r vs oracle %r" % (MODES[mode], n, gc, wc)


@pytest.mark.parametrize("n", [2, 3, 5, 8, 9, 15, 16, 17, 20, 24, 25, 29, 32, 33, 48, 50, 64, 65, 80, 96, 100, 112, 127, 128])
@pytest.mark.parametrize("kappa", [2.75, 1.5])
def test_fast_sigma_counts_exact_values_close(nl, oracle, n, kappa):
    # default dispatch: register-resident sigma kernel (+ generic pass on the NaN
    # borders, + exact kernel for undecidable pixels).  Clip counters must equal the
    # oracle's; values agree to summation-order rounding.
    width, height = 131, 37
    frames = make_frames(n, width, height, seed=300 + n)
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, kappa, kappa, exact=False)
    assert gc == wc, "fast sigma n=%d clip counters %r vs oracle %r" % (n, gc, wc)
    assert close_values(got, want), "fast sigma n=%d: %s" % (n, describe_mismatch(got, want))
    # the rounding difference is far below the tolerance in practice
    ok = ~np.isnan(want) & (want != 0)
    assert np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok])) < 2e-6


@pytest.mark.parametrize("n", [2, 3, 5, 8, 9, 15, 16, 17, 20, 24, 25, 29, 32, 33, 48, 50, 64, 65, 80, 96, 100, 112, 127, 128])
@pytest.mark.parametrize("kappa", [2.75, 1.5])
def test_fast_winsor_counts_exact_values_close(nl, oracle, n, kappa):
    # default dispatch for winsorized sigma clipping: the same register-resident
    # kernel, with the winsorization loop of stack.go:646-672 carried as an interval
    width, height = 131, 37
    frames = make_frames(n, width, height, seed=400 + n, ties=(n % 5 == 0))
    got, gc, want, wc = run_both(nl, oracle, 3, frames, width, height, None, kappa, kappa, exact=False)
    assert gc == wc, "fast winsor n=%d clip counters %r vs oracle %r" % (n, gc, wc)
    assert close_values(got, want), "fast winsor n=%d: %s" % (n, describe_mismatch(got, want))
    ok = ~np.isnan(want) & (want != 0)
    assert np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok])) < 2e-6


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("n", [24, 32, 48, 64, 80, 96, 112, 128])
@pytest.mark.parametrize("nan_frac", [0.0, 0.01, 0.05])
def test_tight_zonal_variant_exact_sizes(nl, oracle, mode, n, nan_frac):
    # a stack of exactly NS frames runs the TIGHT instantiation (no high-zone positions
    # reserved for missing samples): pixels with a few NaNs stay in it with less room to
    # clip, heavier ones and heavy clipping go to the generic pass -- counters exact either way
    width, height = 192, 24
    frames = make_frames(n, width, height, seed=900 + n, nan_frac=nan_frac, hot=0.03, cold=0.01,
                         nan_border=False, all_nan_patch=False)
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        st.set_exact(False)
        got, cl, ch = st.run(mode, 2.5, 2.0, 0.0)
        name = st.last_kernel_name
    assert name == "stack_sigma_fast_kernel<%d, true, %s, true, false, false>" % (n, "true" if mode == 3 else "false"), name
    rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, None, 2.5, 2.0, 0.0, num_cpu=4)
    assert rc == 0
    assert (cl, ch) == (wl, wh), "n=%d mode=%d clip counters %r vs oracle %r" % (n, mode, (cl, ch), (wl, wh))
    assert close_values(got, want), describe_mismatch(got, want)


def test_fast_winsor_clean_frames_and_outliers(nl, oracle):
    # no missing samples (zonal passes only), heavy outliers in a few frames
    width, height, n = 256, 64, 128
    frames = make_frames(n, width, height, seed=19, nan_frac=0.0, nan_border=False,
                         all_nan_patch=False)
    rng = np.random.default_rng(5)
    hot = rng.integers(0, width * height, 4000)
    frames[rng.integers(0, n, 4000), hot] *= 40.0
    frames[:, 100] = 77.0                 # constant pixel: std 0, 0/0 factor, loop ends on changed == 0
    got, gc, want, wc = run_both(nl, oracle, 3, frames, width, height, None, 3.0, 2.0, exact=False)
    assert gc == wc and close_values(got, want)


@pytest.mark.parametrize("n", [2, 3, 8, 9, 16, 31, 32, 47, 64, 65, 100, 128])
def test_fast_median_is_bit_exact(nl, oracle, n):
    # default dispatch for the median: register-resident sorting network; the
    # median is order independent, so it must equal the oracle bit for bit
    width, height = 131, 23
    frames = make_frames(n, width, height, seed=500 + n, ties=(n % 3 == 0))
    frames[0, 7] = np.inf
    frames[n - 1, 9] = -np.inf
    got, _, want, _ = run_both(nl, oracle, 0, frames, width, height, None, 0, 0, exact=False)
    assert same_values(got, want), "fast median n=%d: %s" % (n, describe_mismatch(got, want))


@pytest.mark.parametrize("n", [129, 200, 256, 257, 384, 512])
@pytest.mark.parametrize("clean", [False, True])
def test_multi_lane_winsor_129_to_512_frames(nl, oracle, n, clean):
    # the winsorized variant of the multi-lane kernel (configuration C3: 512 frames)
    width, height = 67, 9
    if clean:
        frames = make_frames(n, width, height, seed=800 + n, nan_frac=0.0, nan_border=False,
                             all_nan_patch=False)
    else:
        frames = make_frames(n, width, height, seed=800 + n, nan_frac=0.01)
    got, gc, want, wc = run_both(nl, oracle, 3, frames, width, height, None, 3.0, 2.5, exact=False)
    assert gc == wc, "multi-lane winsor n=%d clip counters %r vs oracle %r" % (n, gc, wc)
    assert close_values(got, want), "multi-lane winsor n=%d: %s" % (n, describe_mismatch(got, want))


@pytest.mark.parametrize("n", [129, 200, 256, 257, 300, 384, 500, 512])
@pytest.mark.parametrize("clean", [False, True])
def test_multi_lane_sigma_129_to_512_frames(nl, oracle, n, clean):
    # 2 or 4 lanes per pixel (stack_fast_ml.hip): cross-lane merge, quad reductions.
    # clean=True keeps every wave in the zonal passes (no missing samples)
    width, height = 67, 9
    if clean:
        frames = make_frames(n, width, height, seed=700 + n, nan_frac=0.0, nan_border=False,
                             all_nan_patch=False)
    else:
        frames = make_frames(n, width, height, seed=700 + n, nan_frac=0.01)
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, 3.0, 2.5, exact=False)
    assert gc == wc, "multi-lane sigma n=%d clip counters %r vs oracle %r" % (n, gc, wc)
    assert close_values(got, want), "multi-lane sigma n=%d: %s" % (n, describe_mismatch(got, want))


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("n", [129, 144, 145, 200, 249, 252, 256, 257, 300, 384, 385, 470, 505, 509, 512])
@pytest.mark.parametrize("case", ["clean", "nan", "ties", "heavy", "tight"])
def test_lds_column_kernels_129_to_512_frames(nl, oracle, mode, n, case):
    # stack_fast_mlz.hip: clipping / winsorization rounds on LDS columns with walking pointers.
    # "heavy": 6 % hot and 3 % cold outliers -- clips and clamps run to the ends of the columns
    # (hand-over to the generic pass); "tight": kappa 1.2 / 1.0 clips a fifth of the samples;
    # "ties": values quantised to 16 -- runs of equal samples across the pointers
    width, height = 97, 11
    kw = dict(nan_frac=0.0, nan_border=False, all_nan_patch=False)
    sl, sh = 3.0, 2.5
    if case == "nan":
        kw = dict(nan_frac=0.01)
    elif case == "ties":
        kw = dict(nan_frac=0.002, ties=True)
    elif case == "heavy":
        kw = dict(nan_frac=0.0, nan_border=False, all_nan_patch=False, hot=0.06, cold=0.03)
    elif case == "tight":
        sl, sh = 1.2, 1.0
    frames = make_frames(n, width, height, seed=900 + n + 7 * mode, **kw)
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        got, cl, ch = st.run(mode, sl, sh, 0.0)
        assert st.last_kernel_name.startswith("stack_sigma_mlz_kernel<"), st.last_kernel_name
    rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, None, sl, sh, 0.0, num_cpu=4)
    assert rc == 0
    assert (cl, ch) == (wl, wh), "%s n=%d mode %d clip counters %r vs oracle %r" % (case, n, mode, (cl, ch), (wl, wh))
    assert close_values(got, want), "%s n=%d mode %d: %s" % (case, n, mode, describe_mismatch(got, want))


@pytest.mark.parametrize("n", [1, 2, 3, 7, 33, 64, 65, 128, 130, 300, 512])
def test_wave_per_pixel_exact_replay_is_bit_exact(nl, oracle, n):
    # stack_exact_coop.hip: 64 lanes replay ONE pixel in the reference's order
    # (the fallback of the register-resident kernels); forced here for every pixel
    width, height = 37, 5
    frames = make_frames(n, width, height, seed=900 + n, ties=(n % 2 == 1))
    weights = np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32)
    for mode in (2, 3):                   # sigma, winsorized sigma
        for kappa in (2.75, 1.0):
            for w in (None, weights):     # weighted: the weights follow the clip swaps only (stack.go:487)
                got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, w, kappa, kappa, exact=2)
                assert same_values(got, want), "coop %s n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
                assert gc == wc


@pytest.mark.parametrize("n", [1, 2, 3, 7, 15, 16, 17, 33, 64, 65, 128, 130, 300, 512])
def test_four_pixels_per_wave_exact_replay_is_bit_exact(nl, oracle, experiments, n):
    # stack_exact_coop4.hip: the same replay with four pixels per wave on 16-lane rows (row_shr chains); forced
    # for every pixel (37 x 5 = 185 pixels: the last wave holds one pixel, rows with different sample counts,
    # NaN borders, ties)
    width, height = 37, 5
    frames = make_frames(n, width, height, seed=1900 + n, ties=(n % 2 == 1))
    weights = np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32)
    for mode in (2, 3):
        for kappa in (2.75, 1.0):
            for w in (None, weights):
                got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, w, kappa, kappa, exact=4)
                assert same_values(got, want), "coop4 %s n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
                assert gc == wc


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 9, 25, 31, 33, 64, 65, 100, 128])
def test_register_resident_linear_fit_is_bit_exact(nl, oracle, n):
    # default dispatch for linear fit (stack_linfit.hip): one sort, then every sum
    # of stack.go:869-911 runs sequentially in sorted order -> bit-exact
    width, height = 70, 11
    frames = make_frames(n, width, height, seed=600 + n, ties=(n % 4 == 1))
    if n > 3:
        frames[1, 3] = np.inf            # pixels with infinite samples go to the LDS kernel
        frames[2, 40] = -np.inf
    frames[:, 17] = 1234.5               # zero variance: NaN slope, no rejection
    for sl, sh in ((2.75, 2.75), (1.0, 3.0)):
        got, gc, want, wc = run_both(nl, oracle, 5, frames, width, height, None, sl, sh, exact=False)
        assert same_values(got, want), "linear fit n=%d: %s" % (n, describe_mismatch(got, want))
        assert gc == wc, "linear fit n=%d clip counters %r vs oracle %r" % (n, gc, wc)


@pytest.mark.parametrize("n", [24, 57, 96, 120, 127, 128])
@pytest.mark.parametrize("case", ["clean", "nan", "holes", "hot", "tight", "ties"])
def test_linear_fit_chunk_classes_are_bit_exact(nl, oracle, n, case):
    # stack_linfit.hip classifies chunks of 8 column positions per wave and iteration: all alive in
    # every fitting lane (no liveness arithmetic; rejects there are detected from the extremes of
    # the residuals and then the masked pass runs), all dead (skipped), mixed (masked).  Every route
    # must be bit-exact: whole waves without missing samples (clean: all three classes over the
    # iterations), missing samples at the top of the column (nan), ragged columns (holes: 30 % NaN),
    # many rejections (hot: 30 % outliers; tight: kappa 0.8 cuts into the middle), ties.
    width, height = 96, 24
    kw = dict(nan_frac=0.0, nan_border=False, all_nan_patch=False)
    if case == "nan":
        kw = dict(nan_frac=0.03)
    elif case == "holes":
        kw = dict(nan_frac=0.3)
    elif case == "hot":
        kw.update(hot=0.2, cold=0.1)
    frames = make_frames(n, width, height, seed=7100 + n, ties=(case == "ties"), **kw)
    sl, sh = (0.8, 0.8) if case == "tight" else (3.0, 2.5)
    got, gc, want, wc = run_both(nl, oracle, 5, frames, width, height, None, sl, sh, exact=False)
    assert same_values(got, want), "linear fit n=%d %s: %s" % (n, case, describe_mismatch(got, want))
    assert gc == wc, "linear fit n=%d %s clip counters %r vs oracle %r" % (n, case, gc, wc)


def test_fast_sigma_clean_frames_no_nan(nl, oracle):
    # no missing samples at all: every wave stays in the zonal passes
    width, height, n = 256, 64, 128
    frames = make_frames(n, width, height, seed=17, nan_frac=0.0, nan_border=False,
                         all_nan_patch=False)
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, 3.0, 3.0, exact=False)
    assert gc == wc and close_values(got, want)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 33, 64, 65, 127, 128, 130, 256, 300, 420])
def test_tile_exact_replay_is_bit_exact(nl, oracle, n):
    # stack_exact_tile.hip: one wave = 64 consecutive pixels, columns in LDS, one pixel per lane,
    # quickselect as a lock-step state machine; forced here for every pixel (set_exact(3)).
    # 200 px wide: full tiles, a ragged last tile, NaN borders, all-NaN pixels, ties.
    width, height = 200, 3
    frames = make_frames(n, width, height, seed=1900 + n, ties=(n % 2 == 1))
    weights = np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32)
    for mode in (2, 3):                   # sigma, winsorized sigma
        for kappa in (2.75, 1.0):
            for w in (None, weights):     # weighted: the weights follow the clip swaps only (stack.go:487)
                with nl.StackHandle(n, width, height) as st:
                    st.upload_frames(frames)
                    st.set_weights(w)
                    st.set_exact(3)
                    got, cl, ch = st.run(mode, kappa, kappa, 0.0)
                    # (above 256 frames the weight indices are 16-bit: 420 frames still fit the 160 KiB LDS)
                    assert st.last_kernel_name.startswith("stack_sigma_tile_kernel"), st.last_kernel_name
                rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, w, kappa, kappa, 0.0, num_cpu=4)
                assert same_values(got, want), "tile %s n=%d w=%s: %s" % (
                    MODES[mode], n, w is not None, describe_mismatch(got, want))
                assert (cl, ch) == (wl, wh)


def test_tile_replay_heavy_clipping_and_degenerate_bounds(nl, oracle):
    # many clips per pixel (swap-with-last chains, clipped samples arriving from the tail),
    # negative / zero sigmas (inverted bounds: everything clipped), constant pixels (stddev 0)
    width, height, n = 128, 2, 40
    frames = make_frames(n, width, height, seed=77, hot=0.2, cold=0.15, nan_frac=0.1)
    frames[:, 5] = 42.0
    frames[:, 6] = np.where(np.arange(n) % 2 == 0, 1.0, 3.0)
    w = np.random.default_rng(3).uniform(0.2, 1.0, n).astype(np.float32)
    for mode in (2, 3):
        for sl, sh in ((0.5, 0.5), (0.0, 3.0), (-1.0, -1.0), (3.0, 0.25)):
            for weights in (None, w):
                with nl.StackHandle(n, width, height) as st:
                    st.upload_frames(frames)
                    st.set_weights(weights)
                    st.set_exact(3)
                    got, cl, ch = st.run(mode, sl, sh, 7.0)
                rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, weights, sl, sh, 7.0, num_cpu=4)
                assert same_values(got, want), "%s (%g,%g): %s" % (MODES[mode], sl, sh, describe_mismatch(got, want))
                assert (cl, ch) == (wl, wh)


def test_weighted_clip_modes_default_dispatch(nl):
    # each depth runs the replay engine that measured fastest over a whole tile (stack_kernels.h): 64 pixels per
    # wave for shallow stacks, one pixel per wave behind the decision pass (33 ... 512 frames) and beyond (four
    # pixels per wave only where a winsorized stack of 129 ... 176 frames has no decision pass: developer switch 4)
    tile, four, one = "stack_sigma_tile_kernel<", "stack_sigma_coop4_kernel<", "stack_sigma_coop_kernel<"
    for n, sigma, winsor in ((16, tile, tile), (32, tile, tile), (33, tile, one), (40, tile, one), (41, one, one),
                             (96, one, one), (128, one, one), (129, one, one), (176, one, one), (177, one, one), (512, one, one),
                             (300, one, one)):
        with nl.StackHandle(n, 64, 4) as st:
            st.fill_synthetic(1)
            st.set_weights(np.linspace(0.2, 1.0, n).astype(np.float32))
            for mode, prefix in ((2, sigma), (3, winsor)):
                st.run(mode, 2.0, 2.0)
                assert st.last_kernel_name.startswith(prefix), (n, mode, st.last_kernel_name)


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("n", [2, 7, 16, 33, 128])
def test_weighted_modes_match_oracle(nl, oracle, mode, n):
    width, height = 64, 31
    frames = make_frames(n, width, height, seed=200 + n)
    rng = np.random.default_rng(n)
    weights = (0.2 + 0.8 * rng.random(n)).astype(np.float32)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, weights, 2.0, 3.0)
    assert same_values(got, want), "%s weighted n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
    if mode >= 2:
        assert gc == wc


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("n", [64, 100, 127, 128, 129, 200])
@pytest.mark.parametrize("width,height", [(64, 32), (65, 17)])
@pytest.mark.parametrize("engine", [0, 2])          # default dispatch (decision pass + replay up to 128 frames) / the replay alone
def test_weighted_replay_partition_corner_cases(nl, oracle, mode, n, width, height, engine):
    # columns chosen for the register partition passes of the wave-per-pixel replay (stack_exact_coop.hip):
    # descending order (every pair of the first pass swaps: 64 swaps at 128 frames, the one pass that falls back to LDS),
    # ascending order (no swap at all), all samples equal and two-valued columns (every position a candidate of both
    # sides), a saw tooth; 64 x 32 pixels go four to a work item, 65 x 17 (odd) one at a time
    rng = np.random.default_rng(n * 7 + width)
    k = np.arange(n, dtype=np.float32)[:, None]
    px = np.arange(width * height, dtype=np.float32)[None, :]
    base = 1000.0 + 0.25 * px
    frames = (base + rng.normal(0.0, 30.0, (n, width * height))).astype(np.float32)
    npix = width * height
    frames[:, 0::8] = (base - 3.0 * k)[:, 0::8]                       # descending
    frames[:, 1::8] = (base + 3.0 * k)[:, 1::8]                       # ascending
    frames[:, 2::8] = np.broadcast_to(base, (n, npix))[:, 2::8]       # all equal
    frames[:, 3::8] = (base + 50.0 * (k % 2))[:, 3::8]                # two values
    frames[:, 4::8] = (base + 10.0 * (k % 7) - 4.0 * (k % 3))[:, 4::8]   # saw tooth with ties
    frames[5 % n, 5::8] = 30000.0                                     # one hot sample: a second round
    frames = frames.reshape(n, height, width)
    weights = (0.2 + 0.8 * rng.random(n)).astype(np.float32)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, weights, 2.5, 2.5, exact=engine)
    assert same_values(got, want), "%s weighted n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
    assert gc == wc


@pytest.mark.parametrize("mode", [0, 2, 3, 4, 5])
def test_ties_and_asymmetric_sigmas(nl, oracle, mode):
    width, height = 50, 20
    frames = make_frames(40, width, height, seed=77, ties=True)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, None, 1.0, 4.0)
    assert same_values(got, want), describe_mismatch(got, want)
    if mode >= 2:
        assert gc == wc


def test_ref_frame_loc_fills_pixels_without_data(nl, oracle):
    # stack.go:388-397: a pixel that is NaN in every frame gets RefFrameLoc
    width, height = 16, 8
    frames = make_frames(9, width, height, seed=5)
    for mode in range(6):
        got, _, want, _ = run_both(nl, oracle, mode, frames, width, height, None, 2.75, 2.75,
                                   ref_loc=123.5)
        empty = np.isnan(frames).all(axis=0)
        assert empty.any()
        assert np.all(got[empty] == np.float32(123.5))
        assert same_values(got, want)


def test_linear_fit_and_median_ignore_weights(nl, oracle):
    # stack.go:158 and :188-189: weights are computed, then not passed
    width, height = 32, 8
    frames = make_frames(30, width, height, seed=9)
    w = np.linspace(0.2, 1.0, 30).astype(np.float32)
    for mode in (0, 5):
        got_w, c_w, want, wc = run_both(nl, oracle, mode, frames, width, height, w, 2.75, 2.75)
        assert same_values(got_w, want)
        if mode == 5:
            assert c_w == wc


def test_weighted_mad_is_an_error_not_a_panic(nl):
    from nightlight_amd import capi
    with nl.StackHandle(8, 16, 4) as st:
        st.upload_frames(make_frames(8, 16, 4, seed=1))
        st.set_weights(np.ones(8, np.float32))
        with pytest.raises(capi.NlError) as e:
            st.run(4, 2.0, 2.0)
        assert e.value.code == capi.ERR_WEIGHTED_MAD
        assert "MADSigma stacking with weights" in e.value.message


def test_invalid_mode(nl):
    from nightlight_amd import capi
    with nl.StackHandle(4, 8, 4) as st:
        for bad in (-1, 7):
            with pytest.raises(capi.NlError) as e:
                st.run(bad)
            assert e.value.code == capi.ERR_INVALID_MODE
            assert e.value.message == "invalid stacking mode"


def test_auto_mode_selection(nl, oracle):
    # stack.go:45-55
    for n, expect in ((3, 1), (6, 2), (15, 3), (25, 5)):
        frames = make_frames(n, 16, 8, seed=n)
        with nl.StackHandle(n, 16, 8) as st:
            st.upload_frames(frames)
            st.set_exact(True)
            got, cl, ch = st.run(6, 2.75, 2.75)
            assert st.last_mode == expect
        rc, want, wl, wh, mu = oracle.stack_apply(6, frames, None, 2.75, 2.75)
        assert mu == expect and same_values(got, want) and (cl, ch) == (wl, wh)


def test_infinite_samples_are_data(nl, oracle):
    # math.IsNaN is the only filter (quirk Q11): +-Inf samples take part
    width, height = 16, 4
    frames = make_frames(12, width, height, seed=3, nan_frac=0.0, nan_border=False,
                         all_nan_patch=False)
    frames[2, 5] = np.inf
    frames[3, 9] = -np.inf
    for mode in (0, 1, 2, 4, 5):
        got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, None, 2.75, 2.75)
        assert same_values(got, want), "mode %d: %s" % (mode, describe_mismatch(got, want))
        if mode >= 2:
            assert gc == wc


def test_negative_sigma_degenerates_like_the_reference(nl, oracle):
    # quirk Q6: sigma=-1 inverts the bounds, first pass clips everything
    width, height = 32, 4
    frames = make_frames(10, width, height, seed=4)
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, -1.0, -1.0)
    assert same_values(got, want) and gc == wc
    # the register-resident kernel hands inverted bounds to the exact kernel
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, -1.0, -1.0, exact=False)
    assert close_values(got, want) and gc == wc
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, 0.0, 3.0, exact=False)
    assert close_values(got, want) and gc == wc


def test_fast_sigma_hands_infinite_samples_to_the_exact_kernel(nl, oracle):
    width, height = 64, 8
    frames = make_frames(64, width, height, seed=31, nan_frac=0.0, nan_border=False,
                         all_nan_patch=False)
    frames[2, 5] = np.inf
    frames[3, 9] = -np.inf
    frames[7, 100] = np.inf
    frames[8, 100] = np.inf
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, 2.75, 2.75, exact=False)
    assert gc == wc
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert close_values(got[ok], want[ok])


def test_fast_sigma_constant_and_tied_pixels(nl, oracle):
    # zero variance, heavy ties and two-valued pixels: the bound interval collapses
    width, height, n = 64, 4, 64
    frames = np.full((n, width * height), 1000.0, np.float32)
    frames[:, 64:128] = (np.arange(n, dtype=np.float32) % 2)[:, None] * 8 + 500      # two values
    frames[:, 128:192] = np.round(make_frames(n, 64, 1, seed=8, nan_frac=0, nan_border=False,
                                              all_nan_patch=False) / 32) * 32          # ties
    frames[5, 10] = 5000.0                                                             # one outlier
    got, gc, want, wc = run_both(nl, oracle, 2, frames, width, height, None, 2.0, 2.0, exact=False)
    assert gc == wc and close_values(got, want), describe_mismatch(got, want)


def test_tiles_reassemble_the_whole_image(nl, oracle):
    # row-tile sharding: per-tile results and counters add up
    width, height, n = 40, 24, 20
    frames = make_frames(n, width, height, seed=11)
    rc, want, wl, wh, _ = oracle.stack_apply(3, frames, None, 2.5, 2.5)
    out = np.zeros(width * height, np.float32)
    tl = th = 0
    for row0, rows in ((0, 7), (7, 9), (16, 8)):
        with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
            st.upload_frames(frames)
            st.set_exact(True)
            _, cl, ch = st.run(3, 2.5, 2.5, out=out)
            tl += cl
            th += ch
    assert same_values(out, want) and (tl, th) == (wl, wh)
    # same with the register-resident sigma kernel: counters add up exactly
    rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, 2.5, 2.5)
    out = np.zeros(width * height, np.float32)
    tl = th = 0
    for row0, rows in ((0, 7), (7, 9), (16, 8)):
        with nl.StackHandle(n, width, height, row0=row0, rows=rows) as st:
            st.upload_frames(frames)
            _, cl, ch = st.run(2, 2.5, 2.5, out=out)
            tl += cl
            th += ch
    assert close_values(out, want) and (tl, th) == (wl, wh)


def test_synthetic_fill_is_tile_consistent_and_deterministic(nl):
    width, height, n = 48, 36, 11
    with nl.StackHandle(n, width, height) as st:
        st.fill_synthetic(1234)
        whole = np.stack([st.download_tile(i) for i in range(n)])
        st.fill_synthetic(1234)
        again = np.stack([st.download_tile(i) for i in range(n)])
    assert np.array_equal(whole.view(np.uint32), again.view(np.uint32))
    with nl.StackHandle(n, width, height, row0=10, rows=12) as st:
        st.fill_synthetic(1234)
        part = np.stack([st.download_tile(i) for i in range(n)])
    ref = whole.reshape(n, height, width)[:, 10:22, :].reshape(n, -1)
    assert np.array_equal(part.view(np.uint32), ref.view(np.uint32))
    assert np.isnan(whole).all(axis=0).sum() == 64          # the 8x8 patch without data
    assert 0.001 < np.isnan(whole).mean() < 0.5


def test_larger_stack_sigma_and_goal_seek(nl, oracle):
    # a 256x96 stack of 48 frames: sigma clip, then the bisection goal-seek
    width, height, n = 256, 96, 48
    with nl.StackHandle(n, width, height) as st:
        st.fill_synthetic(99)
        frames = np.stack([st.download_tile(i) for i in range(n)])
        rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, 3.0, 3.0, num_cpu=8)
        op, ores, ocl, och, osl, osh = oracle.find_sigmas_bisect(2, frames, 0.5, 0.5, num_cpu=8)
        for exact in (True, False):
            st.set_exact(exact)
            got, cl, ch = st.run(2, 3.0, 3.0)
            assert (cl, ch) == (wl, wh)
            assert (same_values if exact else close_values)(got, want), describe_mismatch(got, want)
            # goal-seek takes the same bisection path because the counters are exact
            out, cl, ch, sl, sh, passes = st.find_sigmas(2, 0.5, 0.5)
            assert (passes, cl, ch) == (op, ocl, och)
            assert (np.float32(sl), np.float32(sh)) == (osl, osh)
            assert (same_values if exact else close_values)(out, ores)


@pytest.mark.parametrize("mode", [0, 2, 3, 5])
@pytest.mark.parametrize("n", [520, 700])
def test_more_than_512_frames_fall_back_to_the_exact_kernels(nl, oracle, mode, n):
    # beyond the register-resident kernels (N > 512; N > 128 for median / linear fit) the
    # default dispatch is the LDS exact kernel: bit-exact
    width, height = 40, 3
    frames = make_frames(n, width, height, seed=1000 + n, nan_frac=0.01)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, None, 2.75, 2.75, exact=False)
    assert same_values(got, want), "%s n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
    if mode >= 2:
        assert gc == wc


def test_goal_seek_winsorized_512_frames_tile(nl, oracle):
    # configuration C3 in miniature: 512 frames, winsorized sigma clip, bisection on the clip
    # percentages (stackfindsigma.go:48-98) -- the multi-lane kernel keeps the counters exact,
    # so the search takes the oracle's path
    width, height, n = 64, 6, 512
    frames = make_frames(n, width, height, seed=77, nan_frac=0.002)
    op, ores, ocl, och, osl, osh = oracle.find_sigmas_bisect(3, frames, 0.5, 0.5, num_cpu=8)
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        out, cl, ch, sl, sh, passes = st.find_sigmas(3, 0.5, 0.5)
    assert (passes, cl, ch) == (op, ocl, och)
    assert (np.float32(sl), np.float32(sh)) == (osl, osh)
    assert close_values(out, ores)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [200, 400, 600])
def test_large_stacks_weighted_and_mad_on_the_exact_kernel(nl, oracle, mode, n):
    # two LDS columns (values + weights / deviations): narrower tiles from 400 frames on
    width, height = 48, 3
    frames = make_frames(n, width, height, seed=2000 + n, nan_frac=0.01)
    weights = None if mode == 4 else np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, weights, 2.5, 3.0, exact=False)
    # MAD up to 512 frames runs on the multi-lane kernel (mean summed in frame order), the rest is bit-exact
    check = close_values if (mode == 4 and n <= 512) else same_values
    assert check(got, want), "%s n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))
    if mode >= 2:
        assert gc == wc


@pytest.mark.parametrize("n", [129, 200, 256, 257, 300, 500, 512])
def test_multi_lane_median_129_to_512_frames(nl, oracle, n):
    # stack_median_ml_kernel: merged column across 2 / 4 lanes, whole-lane rank lookup; bit-exact
    width, height = 67, 9
    frames = make_frames(n, width, height, seed=1200 + n, nan_frac=0.02, ties=(n % 2 == 0))
    frames[0, 5] = np.inf
    frames[n - 1, 6] = -np.inf
    got, _, want, _ = run_both(nl, oracle, 0, frames, width, height, None, 0, 0, exact=False)
    assert same_values(got, want), "multi-lane median n=%d: %s" % (n, describe_mismatch(got, want))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 9, 16, 25, 33, 64, 65, 100, 128])
def test_register_resident_mad_counts_exact_values_close(nl, oracle, n):
    # default dispatch for MAD clipping (stack_mad_fast_kernel): the bounds come from two medians,
    # so the counters are exact; the mean of the survivors is summed in frame order
    width, height = 131, 17
    frames = make_frames(n, width, height, seed=1400 + n, ties=(n % 4 == 0))
    if n > 3:
        frames[0, 11] = np.inf               # one infinite sample: finite median, handled in place
        frames[:, 13] = 5.0                  # constant pixel: MAD 0, nothing clipped
        # (an infinite MEDIAN is not tested against the oracle: Inf - Inf puts NaNs into the
        # deviations and the reference's quickselect then indexes past the array and panics)
    for sl, sh in ((2.75, 2.75), (0.5, 3.0)):
        got, gc, want, wc = run_both(nl, oracle, 4, frames, width, height, None, sl, sh, exact=False)
        assert gc == wc, "mad n=%d clip counters %r vs oracle %r" % (n, gc, wc)
        assert close_values(got, want), "mad n=%d: %s" % (n, describe_mismatch(got, want))


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("n", [70, 125, 128, 200, 300, 512])
def test_pixels_with_one_two_or_three_samples(nl, oracle, mode, n):
    # pixels that keep 1, 2 or 3 valid samples (aligned frames' corners) go to the generic pass on LDS
    # columns (stack_fast_mlg.hip): a single survivor is its own median and clips nothing
    # (found by tests/sweeps/fuzz_parity.py seed 62, case 3720)
    width, height = 64, 3
    frames = make_frames(n, width, height, seed=9300 + n, nan_frac=0.0, nan_border=False, all_nan_patch=False)
    for keep in (1, 2, 3):
        for k, p in enumerate(range(5 * keep, 5 * keep + 4)):
            frames[:, p] = np.nan
            frames[(7 * k + np.arange(keep) * 11) % n, p] = (1136.0, 900.0, 1500.0)[:keep]
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, None, 0.74, 2.08, exact=False)
    assert gc == wc, "%s n=%d clip counters %r vs oracle %r" % (MODES[mode], n, gc, wc)
    assert close_values(got, want), "%s n=%d: %s" % (MODES[mode], n, describe_mismatch(got, want))


@pytest.mark.parametrize("n", [114, 115, 121, 127, 128])
@pytest.mark.parametrize("case", ["clean", "nan", "ties", "hot"])
def test_mad_selection_kernel_114_to_128_frames(nl, oracle, n, case):
    # stack_mad_bitonic_kernel: the deviations of a sorted column are bitonic, the MAD is selected from
    # half-cleaner minima / maxima without a second sort, clip and mean run on the registers; pixels
    # with fewer than 114 samples ("nan": 4 % missing puts some below) take the two-sort kernel through
    # the hand-over list.  Counters exact, mean within summation-order rounding.
    width, height = 160, 12
    kw = dict(nan_frac=0.0, nan_border=False, all_nan_patch=False)
    if case == "nan":
        kw = dict(nan_frac=0.04)
    elif case == "hot":
        kw.update(hot=0.3, cold=0.15)         # medians and MADs deep in the outliers
    frames = make_frames(n, width, height, seed=8200 + n, ties=(case == "ties"), **kw)
    frames[:, 7] = 3.25                       # constant pixel: MAD 0
    if case != "nan":                         # (with missing samples the infinite ones could become the majority: the oracle panics then)
        frames[n // 2 + 1:, 9] = np.inf       # just under half of the samples infinite: median and MAD still finite
    for sl, sh in ((2.75, 2.75), (0.5, 3.0)):
        got, gc, want, wc = run_both(nl, oracle, 4, frames, width, height, None, sl, sh, exact=False)
        assert gc == wc, "mad n=%d %s clip counters %r vs oracle %r" % (n, case, gc, wc)
        assert close_values(got, want), "mad n=%d %s: %s" % (n, case, describe_mismatch(got, want))


@pytest.mark.parametrize("n", [129, 160, 200, 256, 257, 300, 384, 512])
def test_multi_lane_linear_fit_129_to_512_frames(nl, oracle, n):
    # stack_linfit_ml_kernel: the sequential sums are chained through the 2 / 4 lanes of a pixel
    # in sorted order -> bit-exact, counters included
    width, height = 67, 5
    frames = make_frames(n, width, height, seed=1600 + n, nan_frac=0.01, ties=(n % 3 == 0))
    frames[1, 3] = np.inf                # infinite sample -> exact kernel
    frames[:, 17] = 1234.5               # zero variance: NaN slope, no rejection
    for sl, sh in ((2.75, 2.75), (1.0, 3.0)):
        got, gc, want, wc = run_both(nl, oracle, 5, frames, width, height, None, sl, sh, exact=False)
        assert same_values(got, want), "multi-lane linear fit n=%d: %s" % (n, describe_mismatch(got, want))
        assert gc == wc, "multi-lane linear fit n=%d clip counters %r vs oracle %r" % (n, gc, wc)


@pytest.mark.parametrize("n", [129, 200, 256, 257, 300, 512])
def test_multi_lane_mad_129_to_512_frames(nl, oracle, n):
    # stack_mad_ml_kernel: two sort + merge passes and a second read; counters exact
    width, height = 67, 7
    frames = make_frames(n, width, height, seed=1800 + n, nan_frac=0.02, ties=(n % 2 == 1))
    frames[0, 11] = np.inf
    frames[:, 13] = 5.0
    for sl, sh in ((2.75, 2.75), (0.5, 3.0)):
        got, gc, want, wc = run_both(nl, oracle, 4, frames, width, height, None, sl, sh, exact=False)
        assert gc == wc, "multi-lane mad n=%d clip counters %r vs oracle %r" % (n, gc, wc)
        assert close_values(got, want), "multi-lane mad n=%d: %s" % (n, describe_mismatch(got, want))


def test_newton_goal_seek_for_the_linear_fit(nl, oracle):
    # stackfindsigma.go:101-170 (the StLinearFit branch of FindSigmasAndStack): Newton steps from (6, 6)
    # with probe passes at +0.005.  The linear-fit counters are exact, so the device takes the oracle's
    # path pass for pass -- quirks included (high deltas against the LOW target, counter += 3)
    width, height, n = 256, 64, 30
    frames = make_frames(n, width, height, seed=9)
    passes, ores, ocl, och, osl, osh = oracle.find_sigmas_newton(5, frames, 1.0, 1.0, num_cpu=8)
    assert passes == 13                                   # several Newton iterations, not the zero-derivative exit
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        out, cl, ch, sl, sh, got_passes = st.find_sigmas(5, 1.0, 1.0)
    assert (got_passes, cl, ch) == (passes, ocl, och)
    assert (np.float32(sl), np.float32(sh)) == (osl, osh)
    assert close_values(out, ores), describe_mismatch(out, ores)
    # the same through the single-process multi-tile fan-out (host-summed counters)
    with nl.StackGroup(n, width, height, devices=[0, 0, 0]) as g:
        g.upload_frames(frames)
        out, cl, ch, sl, sh, got_passes = g.find_sigmas(5, 1.0, 1.0)
    assert (got_passes, cl, ch, np.float32(sl), np.float32(sh)) == (passes, ocl, och, osl, osh)
    assert close_values(out, ores)
    # zero derivative: a probe pass overwrote the result, which has to be re-made at (6, 6)
    small = make_frames(30, 64, 16, seed=5)
    passes, ores, ocl, och, osl, osh = oracle.find_sigmas_newton(5, small, 1.0, 1.0, num_cpu=2)
    with nl.StackHandle(30, 64, 16) as st:
        st.upload_frames(small)
        out, cl, ch, sl, sh, got_passes = st.find_sigmas(5, 1.0, 1.0)
    assert (got_passes, cl, ch, sl, sh) == (2, ocl, och, 6.0, 6.0) and close_values(out, ores)


def test_goal_seek_on_modes_without_sigmas_stacks_once(nl, oracle):
    # stackfindsigma.go:42-46: "does not support sigmas, proceeding with normal stack" -- Stack(..., 0, 0)
    frames = make_frames(9, 64, 8, seed=3)
    with nl.StackHandle(9, 64, 8) as st:
        st.upload_frames(frames)
        for mode in (0, 1):
            out, cl, ch, sl, sh, passes = st.find_sigmas(mode, 1.0, 1.0)
            rc, want, _, _, _ = oracle.stack_apply(mode, frames, None, 0.0, 0.0)
            assert (passes, sl, sh) == (1, 0.0, 0.0) and same_values(out, want)


@pytest.mark.parametrize("mode,n,weighted", [(2, 128, False), (3, 128, False), (3, 300, False), (2, 512, False),
                                             (2, 100, True), (3, 64, True), (3, 128, True)])
def test_developer_switches_do_not_change_results(nl, oracle, mode, n, weighted):
    # nl_stack_set_dev_flags: plain pass protocol (1), replay in front of the generic pass (2), no decision pass /
    # no recorded rounds (4) -- and a second pass on the same handle (fused protocol, grids sized from the first
    # pass's list lengths): every combination must give the bits and the counters of the default
    width, height = 4096, 12
    with nl.StackHandle(n, width, 4096, row0=0, rows=height) as st:
        st.fill_synthetic(5)
        if weighted:
            st.set_weights(np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32))
        ref = None
        for flags in (0, 0, 1, 2, 4, 7, 0):
            st.set_dev_flags(flags)
            got, cl, ch = st.run(mode, 3.0, 2.5)
            got = got[:height * width]
            if ref is None:
                ref = (got.copy(), cl, ch)
            assert (cl, ch) == ref[1:], (flags, cl, ch, ref[1:])
            assert np.array_equal(got.view(np.uint32), ref[0].view(np.uint32)), flags


@pytest.mark.parametrize("n,weighted,height", [(512, False, 12), (500, False, 9), (512, True, 6)])
def test_split_lds_column_pass_gives_the_bits_of_the_one_kernel_pass(nl, oracle, experiments, n, weighted, height):
    # developer switch 1024: the selected LDS-column kernel (497 ... 512 frames, plain sigma) as a sorting kernel plus a
    # rounds kernel over columns kept in device memory (FastArgs::cols); 2048: as persistent workgroups that loop over
    # blocks of 64 pixels without a barrier -- same code for the rounds, so the same bits, counters and hand-over lists;
    # a ragged last workgroup (height * width not a multiple of 64) included
    width = 4096 if height != 9 else 1000
    with nl.StackHandle(n, width, 4096, row0=0, rows=height) as st:
        st.fill_synthetic(9)
        if weighted:
            st.set_weights(np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32))
        ref = None
        b0 = None
        for flags in (0, 1024, 1024 | 1, 2048, 2048 | 1, 0):
            st.set_dev_flags(flags)
            got, cl, ch = st.run(2, 3.0, 2.5)
            got = got[:height * width]
            if ref is None:
                ref = (got.copy(), cl, ch, st.last_generic_pixels, st.last_fallback_pixels)
                b0 = st.device_bytes
            assert (cl, ch) == ref[1:3], (flags, cl, ch, ref[1:3])
            assert np.array_equal(got.view(np.uint32), ref[0].view(np.uint32)), flags
            if not weighted:
                assert (st.last_generic_pixels, st.last_fallback_pixels) == ref[3:], flags
        assert st.device_bytes >= b0 + 88 * 4 * height * width          # the columns' buffer is on the books
        if not weighted:
            frames = [st.download_tile(i) for i in range(n)]
            rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, 3.0, 2.5)
            assert (wl, wh) == ref[1:3]
            assert np.allclose(ref[0], want[:height * width], rtol=1e-5, atol=0)


def test_device_bytes_reports_the_lazily_allocated_scratch(nl):
    # nl_stack_device_bytes: create-time buffers, then the decision-pass thresholds a weighted clip mode allocates
    # on its first pass (65 bytes per pixel) and keeps until destroy
    n, w, h = 64, 64, 32                # (weighted sigma clipping: decision pass from 41 frames on)
    with nl.StackHandle(n, w, h) as st:
        b0 = st.device_bytes
        assert b0 >= n * w * h * 4 + w * h * 4
        st.fill_synthetic(seed=3)
        st.run(1)
        assert st.device_bytes == b0                       # a mean pass allocates nothing
        st.set_weights(np.linspace(0.2, 1.0, n).astype(np.float32))
        st.run(2, 3.0, 3.0)
        assert st.device_bytes >= b0 + 65 * w * h


def test_destroyed_handles_park_their_buffers_for_the_next_one(nl, oracle):
    # nl_stack_destroy parks the large buffers, the next handle of the same geometry takes them over (stale contents:
    # every pass must overwrite what it reads back); nl_release_cached_memory hands them back to HIP
    import ctypes
    from nightlight_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.nl_release_cached_memory.restype = None
    n, w, h = 24, 1024, 512                    # 48 MiB of frames: above the cache's 1 MiB floor
    lib.nl_release_cached_memory()
    frames = make_frames(n, w, h, seed=4242)
    with nl.StackHandle(n, w, h) as st:
        p0 = st.frames_device_ptr()
        st.upload_frames(frames)
        first, cl0, ch0 = st.run(2, 2.5, 2.5)
    with nl.StackHandle(n, w, h) as st:
        assert st.frames_device_ptr() == p0    # the parked block
        st.upload_frames(frames[::-1].copy())   # other contents in the same memory
        st.run(3, 2.0, 2.0)
        st.upload_frames(frames)
        again, cl1, ch1 = st.run(2, 2.5, 2.5)
    assert (cl0, ch0) == (cl1, ch1) and bits_equal(first, again)
    rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, 2.5, 2.5, 0.0, num_cpu=4)
    assert (cl0, ch0) == (wl, wh) and close_values(first, want)
    lib.nl_release_cached_memory()
    with nl.StackHandle(n, w, h) as st:
        st.upload_frames(frames)
        third, cl2, ch2 = st.run(2, 2.5, 2.5)
    assert (cl2, ch2) == (cl0, ch0) and bits_equal(third, first)


# ---- round 5: extreme magnitudes through the default dispatch -------------------------------------------------
# The clipping passes of the fast kernels take their own moments with v_rcp_f32 / v_sqrt_f32 and widened margins
# (fast_common.hpp, stack_fast_sigma_impl.hpp, stack_fast_mlz_impl.hpp); the margins decide which pixels are
# handed to the bit-exact replay, i.e. whether the clip counters stay the reference's (stats.go:246-261,
# stack.go:404-430).  Subnormal squares (x 1e-36), squares near the flush threshold (x 1e-30), overflowing
# squares (x 1e30) and a constant third of the tile, with ordinary and degenerate kappas.
_EXTREME_SCALES = {"sub36": 1e-36, "sub30": 1e-30, "big30": 1e30, "const3": None}
_EXTREME_KAPPAS = [(3.0, 3.0), (0.01, 20.0), (-1.0, 2.5), (2.5, 0.0)]


def _extreme_case(mode, n, scale, kidx):
    width, height = 97, 5
    seed = 7000 + 131 * mode + 7 * n + kidx
    frames = make_frames(n, width, height, seed=seed, nan_frac=0.02 if n > 8 else 0.0,
                         ties=bool((n + kidx) & 1))
    s = _EXTREME_SCALES[scale]
    if s is None:
        frames[:, : width * height // 3] = np.float32(np.random.default_rng(seed).uniform(-5, 5))
    else:
        frames = (frames * np.float32(s)).astype(np.float32)
    return width, height, frames


@pytest.mark.parametrize("kidx", range(len(_EXTREME_KAPPAS)))
@pytest.mark.parametrize("scale", sorted(_EXTREME_SCALES))
@pytest.mark.parametrize("n", [8, 24, 32, 100, 128, 200, 512])
@pytest.mark.parametrize("mode", [2, 3, 4])
def test_extreme_magnitudes_default_dispatch(nl, oracle, mode, n, scale, kidx):
    sl, sh = _EXTREME_KAPPAS[kidx]
    width, height, frames = _extreme_case(mode, n, scale, kidx)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, None, sl, sh, exact=False)
    assert gc == wc, "%s n=%d %s kappa %r: clip counters %r vs oracle %r" % (MODES[mode], n, scale, (sl, sh), gc, wc)
    assert close_values(got, want), "%s n=%d %s kappa %r: %s" % (MODES[mode], n, scale, (sl, sh),
                                                                   describe_mismatch(got, want))


@pytest.mark.parametrize("scale", ["sub36", "big30"])
@pytest.mark.parametrize("n", [32, 128, 200])
@pytest.mark.parametrize("mode", [2, 3])
def test_extreme_magnitudes_weighted_dispatch(nl, oracle, mode, n, scale):
    # the weighted clip modes run the same pass body record-only (decision pass) in front of the bit-exact replay
    width, height, frames = _extreme_case(mode, n, scale, 0)
    w = np.random.default_rng(n).uniform(0.2, 1.0, n).astype(np.float32)
    got, gc, want, wc = run_both(nl, oracle, mode, frames, width, height, w, 2.75, 2.75, exact=False)
    assert gc == wc, (gc, wc)
    assert same_values(got, want), describe_mismatch(got, want)


# ---- round 5: padded frame stride -------------------------------------------------------------------------------
# Large tiles get a padded frame stride by default (nlstack_api.hip padded_frame_stride); the small images of this
# file stay dense, so the padded layout is forced here through every kernel family: the results must be the BITS of
# the dense layout -- the stride only moves the frames, never the arithmetic -- and the oracle's counters.
@pytest.mark.parametrize("pad", [4, 16448])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("n", [5, 16, 24, 32, 100, 128, 200, 256, 300, 512])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_padded_frame_stride_gives_the_dense_layout_s_bits(nl, oracle, monkeypatch, mode, n, weighted, pad):
    if weighted and mode in (0, 4, 5):
        pytest.skip("mode ignores weights / MADSigma with weights is unimplemented in the reference")
    width, height = 131, 9
    frames = make_frames(n, width, height, seed=9100 + 13 * n + mode, nan_frac=0.02)
    w = np.random.default_rng(n + mode).uniform(0.2, 1.0, n).astype(np.float32) if weighted else None

    def run(stride_pad):
        monkeypatch.setenv("NL_STRIDE_PAD", str(stride_pad))
        with nl.StackHandle(n, width, height) as st:
            assert st.frame_stride() == width * height + stride_pad
            st.upload_frames(frames)
            st.set_weights(w)
            out, cl, ch = st.run(mode, 2.75, 2.75, 0.5)
            back = st.download_tile(n - 1)
            stats = st.frame_stats(n // 2)
            return out, (cl, ch), back, stats

    dense, dc, dback, dstats = run(0)
    padded, pc, pback, pstats = run(pad)
    assert pc == dc and bits_equal(padded, dense), "%s n=%d pad %d: %s" % (MODES[mode], n, pad,
                                                                            describe_mismatch(padded, dense))
    assert bits_equal(pback, frames[n - 1].reshape(pback.shape)) and bits_equal(dback, pback)
    assert np.array_equal(np.asarray(pstats, dtype=np.float64), np.asarray(dstats, dtype=np.float64), equal_nan=True)
    if mode >= 2:
        rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, w, 2.75, 2.75, 0.5, num_cpu=4)
        assert rc == 0 and pc == (wl, wh)


def test_default_frame_stride_and_lent_frames(nl):
    # large tiles: stride = pixels rounded up to 32768 floats + 16448; a second handle borrows the padded buffer with
    # the lender's stride, a dense device buffer with the plain attach; both give the owner's bits
    import torch
    n, width, height = 24, 1024, 300
    frames = make_frames(n, width, height, seed=77, nan_frac=0.01)
    with nl.StackHandle(n, width, height) as st, nl.StackHandle(n, width, height) as h2:
        npix = width * height
        assert st.frame_stride() == (npix + 32767) // 32768 * 32768 + 16448
        assert st.device_bytes >= st.frame_stride() * 4 * n
        st.upload_frames(frames)
        want, wl, wh = st.run(2, 2.5, 2.5)
        h2.attach_device_frames(st.frames_device_ptr(), st.frame_stride())
        assert h2.frame_stride() == st.frame_stride()
        got, cl, ch = h2.run(2, 2.5, 2.5)
        assert (cl, ch) == (wl, wh) and bits_equal(got, want)
        dense = torch.from_numpy(frames.reshape(n, npix)).to("cuda:0").contiguous()
        h2.attach_device_frames(dense.data_ptr())
        assert h2.frame_stride() == npix
        got, cl, ch = h2.run(2, 2.5, 2.5)
        assert (cl, ch) == (wl, wh) and bits_equal(got, want)
        with pytest.raises(Exception):
            h2.attach_device_frames(dense.data_ptr(), npix - 4)
        h2.attach_device_frames(None)
        assert h2.frame_stride() == st.frame_stride()
    with nl.StackHandle(8, 131, 9) as small:
        assert small.frame_stride() == 131 * 9          # small tiles stay dense


# ---- round 5: generic pass and first replay as one launch (stack_tail_fused.hip) ---------------------------------------
# Plain sigma clipping of 65 ... 128 frames with a short exact list runs the tail of a pass as ONE grid (generic pass in the
# lower workgroups, bit-exact replay of the dominant kernel's hand-overs in the upper ones) instead of two streams with a
# join.  The results must be the bits and counters of the two-stream protocol (developer switch 8192) and the oracle's
# counters; queued passes with different kappas -- nothing but stream order between them now -- must end like single ones.
@pytest.mark.parametrize("nan_frac", [0.0, 0.03])
@pytest.mark.parametrize("n", [65, 80, 100, 127, 128])
def test_generic_pass_and_first_replay_in_one_launch(nl, oracle, n, nan_frac):
    width, height = 640, 24
    frames = make_frames(n, width, height, seed=4200 + n, nan_frac=nan_frac, ties=bool(n & 1))
    kappas = [(2.5, 2.5), (2.0, 3.2), (3.0, 2.2)]              # (exact lists of a few dozen pixels: the one-launch tail wants <= 512)
    with nl.StackHandle(n, width, height) as st:
        st.upload_frames(frames)
        st.set_dev_flags(8192)
        st.run(2, *kappas[0])                              # the first pass of a handle leaves the list lengths
        two = [st.run(2, sl, sh) for sl, sh in kappas]
        assert not (st.last_pass_protocol & 2)
        st.set_dev_flags(0)
        one, listed = [], []
        for sl, sh in kappas:
            one.append(st.run(2, sl, sh))
            listed.append(st.last_fallback_pixels)
            assert st.last_pass_protocol & 3 == 3, "pass protocol %d, exact lists %r" % (st.last_pass_protocol, listed)
        assert 0 < max(listed) <= 512
        for (a, al, ah), (b, bl, bh), (sl, sh) in zip(one, two, kappas):
            assert (al, ah) == (bl, bh) and bits_equal(a, b), "kappa %r: %s" % ((sl, sh), describe_mismatch(a, b))
            rc, want, wl, wh, _ = oracle.stack_apply(2, frames, None, sl, sh, 0.0, num_cpu=4)
            assert rc == 0 and (al, ah) == (wl, wh) and close_values(a, want)
        # queued: three passes of different kappas back to back, then the same order again; the last one's result stands
        for sl, sh in kappas + kappas:
            st.run_async(2, sl, sh, 0.0)
        queued = np.zeros(width * height, np.float32)
        cl, ch = st.finish(queued)
        assert st.last_pass_protocol & 2
        assert (cl, ch) == one[-1][1:] and bits_equal(queued, one[-1][0])
        seen = set()
        for _ in range(25):
            out, cl, ch = st.run(2, *kappas[1])
            seen.add((cl, ch, out.tobytes()))
        assert len(seen) == 1


# ---- round 6 ------------------------------------------------------------------------------------------------------------
def test_owned_frame_buffer_is_padded_and_the_stride_is_what_producers_must_use(nl):
    # nlstack 0.2.0: frame k of the OWNED buffer starts at frames_device_ptr + k * frame_stride floats, and for the headline
    # geometry (and every tile of 256 Ki pixels or more) that stride is NOT rows * width.  A producer
    # written against the dense layout of 0.1.0 would corrupt the stack silently: this pins the contract.
    import ctypes
    from nightlight_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.nl_version.restype = ctypes.c_char_p
    assert b"0.2." in lib.nl_version()
    with nl.StackHandle(4, 4096, 4096) as st:                       # the headline's tile (4 frames: 256 MiB)
        npix = 4096 * 4096
        assert st.frame_stride() > npix and st.frame_stride() % 64 == 0, st.frame_stride()
        st.fill_synthetic(3)
        a = st.download_tile(1)
        # the same frame through the raw pointer and the stride (what an in-place producer addresses): a second handle that
        # borrows the buffer with that stride reads the same bits; borrowed as a DENSE buffer it reads frame 1 from elsewhere
        with nl.StackHandle(4, 4096, 4096) as b:
            b.attach_device_frames(st.frames_device_ptr(), st.frame_stride())
            assert bits_equal(b.download_tile(1), a)
            b.attach_device_frames(st.frames_device_ptr())
            assert not bits_equal(b.download_tile(1), a)
            b.attach_device_frames(None)
    with nl.StackHandle(4, 256, 64) as st:                          # small tiles stay dense
        assert st.frame_stride() == 256 * 64


def test_switches_of_the_experiments_build_are_rejected_by_the_default_library(nl):
    # nl_stack_set_exact(h, 4) and developer switches 1024 / 2048 select kernels only libnlstack_exp.so carries: the default
    # library must say so instead of timing its one kernel under another name
    import ctypes
    from nightlight_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.nl_version.restype = ctypes.c_char_p
    exp = b"+experiments" in lib.nl_version()
    with nl.StackHandle(8, 64, 8) as st:
        for call, arg in ((st.set_exact, 4), (st.set_dev_flags, 1024), (st.set_dev_flags, 2048 | 1)):
            if exp:
                call(arg)
            else:
                with pytest.raises(capi.NlError) as e:
                    call(arg)
                assert "experiments build" in str(e.value)
        with pytest.raises(capi.NlError):
            st.set_exact(5)
        st.set_exact(0)
        st.set_dev_flags(0)


def _adversarial_winsor_columns(n, pixels, seed):
    """Columns that sit on the edges of the invariant-interval certificate (stack_fast_sigma_impl.hpp, CERT): ties exactly
    at the clamp, all-equal columns, two-valued columns ({x, y, y}: > 60 winsorization rounds in the reference), one far
    outlier, a standard deviation that RISES in round 1, half the column NaN, and plain noise in between."""
    rng = np.random.default_rng(seed)
    f = (1000.0 + 30.0 * rng.standard_normal((n, pixels))).astype(np.float32)
    k = np.arange(pixels)
    kind = k % 8
    for p in np.flatnonzero(kind == 1):                      # all equal
        f[:, p] = np.float32(1000.0 + (p % 5))
    for p in np.flatnonzero(kind == 2):                      # {x, y, y, ...}
        f[:, p] = np.float32(1200.0)
        f[p % n, p] = np.float32(1000.0)
    for p in np.flatnonzero(kind == 3):                      # ties at median -+ 1.5 sigma of the first round
        col = np.float32(1000.0) + np.float32(8.0) * np.round(rng.standard_normal(n) * 2.0).astype(np.float32)
        f[:, p] = col
    for p in np.flatnonzero(kind == 4):                      # one far outlier: sigma collapses after the first clamp
        f[(p // 8) % n, p] = np.float32(60000.0)
    for p in np.flatnonzero(kind == 5):                      # two clusters: the clamped deviation rises, then falls
        f[: n // 2, p] = np.float32(900.0) + np.float32(0.5) * rng.standard_normal(n // 2).astype(np.float32)
        f[n // 2:, p] = np.float32(1100.0) + np.float32(0.5) * rng.standard_normal(n - n // 2).astype(np.float32)
    for p in np.flatnonzero(kind == 6):                      # half the column missing
        f[rng.permutation(n)[: n // 2], p] = np.nan
    return f


@pytest.mark.parametrize("n", [12, 13, 15, 16, 17, 20, 24, 25, 32, 40, 48, 64, 96, 100, 128])
def test_winsor_certificate_on_and_off_give_identical_counters(nl, oracle, n):
    # the certificate leaves the winsorization loop early on the strength of a monotonicity argument and ad hoc
    # margins; the fuzz logs covered it, the suite did not.  Adversarial columns, certificate on (default) and off (developer
    # switch 16384), negative and asymmetric sigmas included: same counters, values within 1e-5, and the oracle's counters.
    width, height = 512, 6
    frames = _adversarial_winsor_columns(n, width * height, seed=6000 + n)
    for sl, sh in ((3.0, 3.0), (1.0, 2.5), (2.75, 0.5), (-1.0, 2.0)):
        with nl.StackHandle(n, width, height) as st:
            st.upload_frames(frames)
            res = {}
            for flags in (0, 16384, 0):
                st.set_dev_flags(flags)
                res.setdefault(flags, []).append(st.run(3, sl, sh))
            (a, al, ah), (a2, al2, ah2) = res[0]
            (b, bl, bh), = res[16384]
            assert (al, ah) == (bl, bh) == (al2, ah2), "n=%d sigma %r: counters %r / %r / %r" % (n, (sl, sh), (al, ah), (bl, bh), (al2, ah2))
            # (values: a pixel that leaves the loop early is finished by the register kernel, one that runs into the round
            # cap without the certificate by the bit-exact replay -- two summation orders, so 1e-5 and not bits)
            assert close_values(a, b) and close_values(a, a2), "n=%d sigma %r: %s" % (n, (sl, sh), describe_mismatch(a, b))
        rc, want, wl, wh, _ = oracle.stack_apply(3, frames, None, sl, sh, 0.0, num_cpu=4)
        assert rc == 0 and (al, ah) == (wl, wh), "n=%d sigma %r: counters %r vs oracle %r" % (n, (sl, sh), (al, ah), (wl, wh))
        assert close_values(a, want), "n=%d sigma %r: %s" % (n, (sl, sh), describe_mismatch(a, want))
