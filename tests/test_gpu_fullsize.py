"""Full-size parity: one test per BASELINE.json configuration (and the north
star's 512-frame target) at the sizes the numbers are quoted on -- device
buffers of 4 ... 32 GiB, i.e. past 2^31 bytes, where 32-bit offsets, buffer
descriptor ranges and list capacities matter.

The oracle cannot run a whole 4096x4096 stack in seconds, so each test combines
(a) the oracle on randomly chosen rows of the big stack (pixels are independent
    -- internal/ops/stack/stack.go:142-152 -- so this is exact for those rows);
(b) a size-independent property: the tile is re-stacked as a partition of row
    strips, each strip its own small handle filled with the same synthetic
    pixels (the generator is keyed by image coordinates): strip results must be
    bit-identical to the big result's rows and the strip clip counters must sum
    to the big pass's counters (checksum of checksums);
(c) the oracle's clip counters on one strip, which pins (b)'s counters.
Bars as everywhere: counters equal; values bit-exact, or within the north
star's 1e-5 where the register-resident kernels sum in sorted order.
"""
import numpy as np
import pytest

from util import bits_equal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]

RTOL = 1e-5
SEED = 0x4E4C5354


def rel_err(got, want):
    ok = ~np.isnan(want) & (want != got)
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok].astype(np.float64))))


def check_config(nl, oracle, n, width, height, row0, rows, mode, kappa, bit_exact, strip_rows, kernel_prefix,
                 sample_blocks=16, block_rows=4, oracle_strip_rows=8, weights=None, cores=64):
    rng = np.random.default_rng(n * 1000 + mode)
    ow = None if mode in (0, 5) else weights
    with nl.StackHandle(n, width, height, row0=row0, rows=rows) as big:
        big.fill_synthetic(SEED)
        big.set_weights(weights)
        big.run_async(mode, kappa, kappa, 0.0)
        big_cl, big_ch = big.finish()
        kernel = big.last_kernel_name
        assert kernel.startswith(kernel_prefix), kernel          # the fast path ran, not a fallback
        result = big.download_rows(-1, 0, rows)

        # (a) oracle on sampled row blocks (first and last rows always included)
        starts = sorted(set([0, rows - block_rows] +
                            [int(r) for r in rng.integers(0, rows - block_rows + 1, sample_blocks - 2)]))
        worst = 0.0
        for r in starts:
            frames = np.empty((n, block_rows * width), np.float32)
            for k in range(n):
                frames[k] = big.download_rows(k, r, block_rows)
            rc, want, _, _, _ = oracle.stack_apply(mode, frames, ow, kappa, kappa, 0.0, num_cpu=cores)
            assert rc == 0
            got = result[r * width:(r + block_rows) * width]
            assert np.array_equal(np.isnan(got), np.isnan(want)), "rows %d..: NaN pattern differs" % r
            if bit_exact:
                assert bits_equal(got, want), "rows %d..%d: %d pixels differ from the oracle" % (
                    r, r + block_rows, np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
            else:
                worst = max(worst, rel_err(got, want))
        assert worst <= RTOL, "max relative difference %g" % worst

    # (b) partition into strips, each its own handle with the same synthetic pixels
    sum_cl = sum_ch = 0
    for s0 in range(0, rows, strip_rows):
        sr = min(strip_rows, rows - s0)
        with nl.StackHandle(n, width, height, row0=row0 + s0, rows=sr) as strip:
            strip.fill_synthetic(SEED)
            strip.set_weights(weights)
            strip.run_async(mode, kappa, kappa, 0.0)
            cl, ch = strip.finish()
            sres = strip.download_rows(-1, 0, sr)
        sum_cl += cl
        sum_ch += ch
        assert bits_equal(sres, result[s0 * width:(s0 + sr) * width]), "strip at row %d differs from the big pass" % s0
    if mode >= 2:
        assert (sum_cl, sum_ch) == (big_cl, big_ch), "strip counters %r vs big pass %r" % ((sum_cl, sum_ch), (big_cl, big_ch))

    # (c) oracle counters on one small strip (its own handle: the same kernels, small offsets)
    if mode >= 2:
        r = int(rng.integers(0, rows - oracle_strip_rows + 1))
        with nl.StackHandle(n, width, height, row0=row0 + r, rows=oracle_strip_rows) as strip:
            strip.fill_synthetic(SEED)
            strip.set_weights(weights)
            frames = np.stack([strip.download_tile(k) for k in range(n)])
            strip.run_async(mode, kappa, kappa, 0.0)
            cl, ch = strip.finish()
        rc, want, wl, wh, _ = oracle.stack_apply(mode, frames, ow, kappa, kappa, 0.0, num_cpu=cores)
        assert rc == 0 and (cl, ch) == (wl, wh), "strip at row %d: counters %r vs oracle %r" % (r, (cl, ch), (wl, wh))
        assert big_cl + big_ch > 0
    return big_cl, big_ch


def test_c2_sigma_128x4096x4096(nl, oracle):
    # BASELINE.json configs[1]; 8 GiB of frames
    cl, ch = check_config(nl, oracle, 128, 4096, 4096, 0, 4096, mode=2, kappa=3.0, bit_exact=False,
                          strip_rows=1024, kernel_prefix="stack_sigma_fast_kernel<128")
    assert (cl, ch) == (6836157, 13270993)       # the counters every bench line of this stack reports


def test_c3_tile_winsor_512x512x4096(nl, oracle):
    # BASELINE.json configs[2], one GPU's share: rows [1536, 2048) of 512 frames of 4096x4096; 4 GiB
    check_config(nl, oracle, 512, 4096, 4096, 1536, 512, mode=3, kappa=2.75, bit_exact=False,
                 strip_rows=128, kernel_prefix="stack_sigma_mlz_kernel<4", sample_blocks=8, oracle_strip_rows=4)


def test_c4_linear_fit_128x4096x4096_noise_weighted(nl, oracle):
    # BASELINE.json configs[3]: linear-fit rejection with inverse-noise weighting selected.  The
    # reference's linear fit ignores the weights (stack.go:188-189), so they must not change a bit.
    n = 128
    w = (0.2 + 0.8 * ((np.arange(n) * 37) % 101) / 100.0).astype(np.float32)
    check_config(nl, oracle, n, 4096, 4096, 0, 4096, mode=5, kappa=2.75, bit_exact=False,
                 strip_rows=1024, kernel_prefix="stack_linfit", weights=w, sample_blocks=8)


def test_c5_median_64x6000x4000(nl, oracle):
    # BASELINE.json configs[4]; 5.7 GiB; order independent => bit-exact
    check_config(nl, oracle, 64, 6000, 4000, 0, 4000, mode=0, kappa=0.0, bit_exact=True,
                 strip_rows=1000, kernel_prefix="stack_median_fast_kernel<64")


def test_sigma_512x4096x4096(nl, oracle):
    # the north star's roofline target configuration; 32 GiB of frames in one buffer
    check_config(nl, oracle, 512, 4096, 4096, 0, 4096, mode=2, kappa=3.0, bit_exact=False,
                 strip_rows=1024, kernel_prefix="stack_sigma_mlz_kernel<4", sample_blocks=8, oracle_strip_rows=4)


def test_weighted_sigma_128x4096x4096(nl, oracle):
    # the weighted clip modes replay the reference's permutation: bit-exact at full size
    n = 128
    w = (0.2 + 0.8 * ((np.arange(n) * 37) % 101) / 100.0).astype(np.float32)
    check_config(nl, oracle, n, 4096, 4096, 0, 4096, mode=2, kappa=3.0, bit_exact=True,
                 strip_rows=2048, kernel_prefix="stack_", weights=w, sample_blocks=8)


def test_tile_beyond_32bit_offsets_takes_the_int64_kernels(nl, oracle):
    # 2^27 pixels per frame: the one-lane register kernels (32-bit buffer offsets over 4 frames)
    # must step aside; result still equal to the oracle on sampled rows, first and last included
    n, width, height = 3, 16384, 8192
    rng = np.random.default_rng(5)
    with nl.StackHandle(n, width, height) as st:
        st.fill_synthetic(SEED)
        st.run_async(2, 1.5, 1.5, 0.0)
        cl, ch = st.finish()
        assert not st.last_kernel_name.startswith("stack_sigma_fast"), st.last_kernel_name
        for r in [0, height - 2] + [int(x) for x in rng.integers(0, height - 2, 6)]:
            frames = np.stack([st.download_rows(k, r, 2) for k in range(n)])
            rc, want, _, _, _ = oracle.stack_apply(2, frames, None, 1.5, 1.5, 0.0, num_cpu=8)
            assert bits_equal(st.download_rows(-1, r, 2), want), "rows %d.." % r


def _bisection(run_pass, total, perc_lo, perc_hi):
    """binarySearchAndStack (internal/ops/stack/stackfindsigma.go:48-98) restated on the test side
    in fp32; run_pass(sigma_low, sigma_high) -> (clip_low, clip_high).  Returns the trajectory."""
    F = np.float32
    lo_l, lo_r, hi_l, hi_r = F(1), F(11), F(1), F(11)
    lo_m, hi_m = F(0.5) * (lo_l + lo_r), F(0.5) * (hi_l + hi_r)
    steps = []
    i = 0
    while True:
        cl, ch = run_pass(float(lo_m), float(hi_m))
        steps.append((float(lo_m), float(hi_m), cl, ch))
        pl = F(cl) * F(100) / F(total)
        ph = F(ch) * F(100) / F(total)
        dl = int(F(100) * pl + F(0.5)) - int(F(100) * F(perc_lo))
        dh = int(F(100) * ph + F(0.5)) - int(F(100) * F(perc_hi))
        if (dl == 0 and dh == 0) or i >= 20:
            return steps
        if dl > 0:
            lo_l = lo_m
        elif dl < 0:
            lo_r = lo_m
        lo_m = F(0.5) * (lo_l + lo_r)
        if dh > 0:
            hi_l = hi_m
        elif dh < 0:
            hi_r = hi_m
        hi_m = F(0.5) * (hi_l + hi_r)
        i += 1


def test_c3_as_stated_winsor_goal_seek_512x4096x4096_over_8_tiles(nl, oracle):
    """BASELINE.json configs[2] in its stated form: 512 frames of 4096x4096, winsorized sigma
    clipping with the goal-seek on the clip percentages (stackfindsigma.go:48-98, targets 0.5 % /
    0.5 %, README.md:155-156), the image split into 8 row tiles whose counters are summed after
    every pass (stack.go:142-152, 193-198).  One device holds all 8 tiles (8 x 4 GiB) -- the same
    nl_group_* code that puts tile t on GPU t.  Checked:
    (1) the group's goal-seek = the reference's bisection (restated above) driven by the summed
        per-tile counters of every pass: same number of passes, same sigmas, same counters;
    (2) per pass, the group's counters = the sum over 8 FRESH single-tile handles (first and last pass);
    (3) per pass of the trajectory, the counters of two 2-row strips = the oracle's on the same pixels;
    (4) the result at the converged sigmas = the oracle on sampled row blocks, within 1e-5."""
    n, width, height, tiles, mode = 512, 4096, 4096, 8, 3
    total = n * width * height
    rng = np.random.default_rng(33)
    with nl.StackGroup(n, width, height, devices=[0] * tiles) as g:
        assert g.size == tiles and [g.tile_rows(t) for t in range(tiles)] == [(512 * t, 512) for t in range(tiles)]
        g.fill_synthetic(SEED)
        result, cl, ch, sl, sh, passes = g.find_sigmas(mode, 0.5, 0.5)
        views = [g.tile(t) for t in range(tiles)]
        assert all(v.last_kernel_name.startswith("stack_sigma_mlz_kernel<4") for v in views), views[0].last_kernel_name

        # (1) replay the bisection on the per-tile counters of the same tiles
        def group_pass(sig_lo, sig_hi):
            for v in views:
                v.run_async(mode, sig_lo, sig_hi, 0.0)
            per_tile = [v.finish() for v in views]
            return sum(c[0] for c in per_tile), sum(c[1] for c in per_tile)
        steps = _bisection(group_pass, total, 0.5, 0.5)
        assert len(steps) == passes, (len(steps), passes)
        assert steps[-1] == (sl, sh, cl, ch), (steps[-1], (sl, sh, cl, ch))
        assert 2 <= passes <= 21
        # the goal was met: both percentages round to 0.50 (stackfindsigma.go:64-70)
        if passes <= 20:
            assert int(np.float32(100) * (np.float32(cl) * np.float32(100) / np.float32(total)) + np.float32(0.5)) == 50
            assert int(np.float32(100) * (np.float32(ch) * np.float32(100) / np.float32(total)) + np.float32(0.5)) == 50

        # (4) oracle values at the converged sigmas on sampled row blocks (tile seams included)
        block_rows = 2
        starts = sorted(set([0, 510, 512, height - block_rows] +
                            [int(r) for r in rng.integers(0, height - block_rows + 1, 4)]))
        worst = 0.0
        for r in starts:
            t = r // 512
            frames = np.stack([views[t].download_rows(k, r - 512 * t, block_rows) for k in range(n)])
            rc, want, _, _, _ = oracle.stack_apply(mode, frames, None, sl, sh, 0.0, num_cpu=64)
            assert rc == 0
            got = result[r * width:(r + block_rows) * width]
            assert np.array_equal(np.isnan(got), np.isnan(want)), "rows %d..: NaN pattern differs" % r
            worst = max(worst, rel_err(got, want))
        assert worst <= RTOL, "max relative difference %g" % worst
        strip_frames = {}
        for r in (0, 2046):                   # NaN-border rows of tile 0, interior rows at a tile seam
            t = r // 512
            strip_frames[r] = np.stack([views[t].download_rows(k, r - 512 * t, 2) for k in range(n)])

    # (2) fresh single-tile handles: first and last pass of the trajectory
    for (s_lo, s_hi, want_cl, want_ch) in (steps[0], steps[-1]):
        sum_cl = sum_ch = 0
        for t in range(tiles):
            with nl.StackHandle(n, width, height, row0=512 * t, rows=512) as st:
                st.fill_synthetic(SEED)
                st.run_async(mode, s_lo, s_hi, 0.0)
                c = st.finish()
                if (s_lo, s_hi) == (sl, sh):
                    assert bits_equal(st.download_rows(-1, 0, 512), result[512 * t * width:512 * (t + 1) * width])
            sum_cl += c[0]
            sum_ch += c[1]
        assert (sum_cl, sum_ch) == (want_cl, want_ch), ((sum_cl, sum_ch), (want_cl, want_ch))

    # (3) every pass of the trajectory: strip counters against the oracle
    for r, frames in strip_frames.items():
        with nl.StackHandle(n, width, height, row0=r, rows=2) as st:
            st.fill_synthetic(SEED)
            assert bits_equal(st.download_tile(7), frames[7])
            for (s_lo, s_hi, _, _) in steps:
                st.run_async(mode, s_lo, s_hi, 0.0)
                c = st.finish()
                rc, _, wl, wh, _ = oracle.stack_apply(mode, frames, None, s_lo, s_hi, 0.0, num_cpu=64)
                assert rc == 0 and c == (wl, wh), "rows %d.. at sigmas %r: %r vs oracle %r" % (r, (s_lo, s_hi), c, (wl, wh))
