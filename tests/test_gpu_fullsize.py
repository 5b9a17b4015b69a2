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
