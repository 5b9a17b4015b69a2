// stack_fast_sigma_impl.hpp -- the one-lane register-resident sigma / winsorized sigma kernel (see stack_fast.hip for
// the exactness contract).  Included by stack_fast.hip (the stacking passes) and stack_fast_decide.hip (RECORD: the
// same kernel as the DECISION pass of weighted stacks).  The includer defines NL_STAT.
#pragma once
#include "fast_ml_common.hpp"

namespace nl {

// ZONAL = true : grid covers the tile, lane = pixel blockIdx*256+thread;
// ZONAL = false: grid-stride over q.in_list (pixels handed over by the zonal
//                kernel), any number of missing / clipped samples.
// TIGHT (zonal only): the stack has exactly NS frames, so the only missing samples are a
//                pixel's own NaNs -- the high zone reserves no positions for them (a third
//                fewer zone positions to mask, count and re-sum every clipping round); a lane
//                whose NaNs leave no survivor in the high zone goes to the generic pass as before.
// RECORD (zonal only): the DECISION pass of a weighted stack (stack.go:442-531, 710-829).  The rejection of the
//                weighted modes is the unweighted one -- median and standard deviation ignore the weights -- so this
//                kernel takes every clip decision exactly as for an unweighted stack, and instead of a result it
//                leaves, per pixel and clipping round, thresholds (lo, hi) that reproduce the reference's decisions
//                (x < lo, x > hi: no surviving sample lies between the two ends of either bound's interval) in
//                p.bounds[round][pixel], and the number of rounds in p.nrounds[pixel] -- 0 where it could not decide
//                (undecidable sample, zone overflow, too many missing samples or rounds).  The bit-exact replay then
//                only permutes and clips (no sums, no winsorization) and forms the weighted mean.  Writes no result,
//                no counters and no lists.
// The winsorization CASCADE (winsorized zonal kernels; FastArgs::pass_budget, round_cap, cont_*).  A wave runs its
// winsorization loop (stack.go:649-672 as an interval, WinsorInterval) until its slowest lane is through, in every
// clipping pass, and clipping passes until its slowest lane needs no more: measured on the bench stack
// (tools/round_stats.py) a wave executes 44 rounds in 3.3 passes at 16 frames, 28 in 3.4 at 128, where a lane needs 7 ... 10
// rounds in 1.3 ... 1.8 passes -- the kernel's time is those rounds.  A stage of the cascade runs at most pass_budget
// clipping passes per wave with at most round_cap winsorization rounds each: lanes still inside a winsorization loop at
// the cap fall back to the state they entered that pass with, and whoever is not done at the end is appended -- pixel,
// clip counts so far -- to the continuation list.  The next stage (CONT: zonal kernel over that list) gathers and sorts
// those pixels again (deterministic: the counts address the same samples), in freshly packed waves, and carries on.
// Only the stage that finishes a pixel books its counters, so a pixel that turns undecidable in a later stage is
// replayed from scratch as before.
// CONT (zonal only): the kernel runs over q.in_list / q.in_state instead of the tile.

#ifndef NL_CERT_ON
#define NL_CERT_ON(ns) true
#endif
#ifndef NL_WINSOR_WL
#define NL_WINSOR_WL(ns) ((ns) >= 112 ? 20 : ((ns) >= 80 ? 16 : ((ns) / 4 + 3) / 4 * 4))
#endif

// Occupancy.  The winsorized zonal kernels are chains of dependent operations (interval arithmetic, square roots,
// divisions): a third wave per SIMD is worth a few spilled registers.  The instantiations for stacks of exactly NS frames
// fit 168 registers anyway (158 / 166 at 112 / 128 positions); the ones with padding (frame counts between the network
// sizes) took 178 and ran at two waves per SIMD -- 120 frames 8.1 ms where 128 frames take 5.4.
// The same one step down: 48 ... 96 positions take 118 ... 134 registers for exactly NS frames (133 with padding at 48) -- a
// fourth wave costs a handful of spills: 64 / 80 / 96 frames dominant kernel -11 / -8 / -6 %, 44 frames -15 %; the padded
// instantiations of 64 ... 96 positions (155 registers) lose 5 % when forced and stay.
template <int NS, bool ZONAL, bool WINSOR, bool TIGHT, bool RECORD = false, bool CONT = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu((ZONAL && WINSOR && !CONT) ? (NS >= 112 ? (TIGHT ? 1 : 3) : (RECORD ? 1 : (NS >= 64 ? (TIGHT ? 4 : 1) : (NS >= 48 ? 4 : 1)))) : 1, 8)))
void stack_sigma_fast_kernel(StackArgs p, FastArgs q)
{
    static_assert(!TIGHT || ZONAL, "TIGHT is a variant of the zonal kernels");
    static_assert(!RECORD || ZONAL, "RECORD is a variant of the zonal kernels");
    static_assert(!CONT || (ZONAL && WINSOR && !RECORD), "CONT continues a winsorized zonal pass");
    constexpr bool CASCADE = ZONAL && WINSOR && !RECORD;      // this instantiation knows about budgets and continuation lists
    if constexpr (ZONAL && !RECORD && !CONT) fused_prologue_dominant(p);
    if constexpr (!ZONAL) { if (q.in_list) { fused_collect_slots(p, blockIdx.x); snapshot_fb_list(q); } }
    // zone widths: 8 clipped + 8 missing samples per lane for the larger
    // networks, 4 + 4 for the small ones
    // (zones of 8 for the winsorized 24- and 32-position kernels were measured: 553 k -> 86 k pixels in the generic pass
    // at 17 frames, but every round masks twice the positions: pass 3.87 -> 4.19 ms)
    constexpr int KZ = NS >= 48 ? kZone : 4, KP = TIGHT ? 0 : (NS >= 48 ? kPadMax : 4);
    static_assert(!ZONAL || NS >= 16, "zonal passes need room between the zones");
    constexpr int ZL = KZ;                                    // low zone  = positions [0, ZL)
    constexpr int ZH = ZONAL ? NS - KZ - KP : NS;             // high zone = positions [ZH, NS)

    int c_lo_total = 0, c_hi_total = 0;
    // ZONAL, or GENERIC without a list: the grid covers the tile, one pixel per
    // lane, a single trip.  GENERIC with a list: grid-stride over the hand-over
    // list, whose length is only known on the device.
    const bool listed = (!ZONAL || CONT) && q.in_list != nullptr;
    int64_t limit = listed ? (int64_t)min(*q.in_count, q.in_capacity) : p.npix;
    int64_t sweep = listed ? (int64_t)gridDim.x * blockDim.x : limit;
    int64_t wg_first = (int64_t)blockIdx.x * blockDim.x;
    // CASCADE: this workgroup's continuation region fills through an LDS counter (no device atomics, see FastArgs)
    __shared__ unsigned s_cont, s_pref[17], s_gen, s_gen_base;
    if constexpr (CASCADE) {
        if (threadIdx.x == 0) s_cont = 0u;
    }
    if constexpr (ZONAL && !RECORD) {
        if (threadIdx.x == 0) s_gen = 0u;
        __syncthreads();
    }
    if constexpr (CONT) {
        // input: q.in_group consecutive regions of the previous stage's list, first region blockIdx.x * in_group; the
        // items of the group are numbered through (prefix sums of the regions' lengths in s_pref), 256 per trip
        if (threadIdx.x == 0) {
            unsigned acc = 0;
            for (unsigned r = 0; r < q.in_group; r++) {
                const unsigned reg = blockIdx.x * q.in_group + r;
                s_pref[r] = acc;
                acc += reg < q.in_regions ? min(q.in_count[reg], q.in_region) : 0u;
            }
            s_pref[q.in_group] = acc;
        }
        __syncthreads();
        limit = (int64_t)s_pref[q.in_group];
        wg_first = 0;
        sweep = blockDim.x;
    } else if constexpr (CASCADE) {
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;

    for (int64_t wg_item = wg_first; wg_item < limit; wg_item += sweep) {
        // the frame count is re-read through an opaque register every trip:
        // otherwise the compiler hoists the 128 per-frame scalar selects that
        // depend on it out of the loop and spills them
        int N = p.n_frames;
        asm volatile("" : "+s"(N));
        const int64_t item = wg_item + threadIdx.x;
        bool on = item < limit;
        int64_t pix = item;
        unsigned cas_at = 0;                                 // CONT: where this lane's item sits in the previous stage's list
        if constexpr (CONT) {
            unsigned r = 0;
            for (unsigned k = 1; k < q.in_group; k++) r += ((unsigned)item >= s_pref[k]) ? 1u : 0u;      // (regions may be empty: count them all)
            cas_at = (blockIdx.x * q.in_group + r) * q.in_region + ((unsigned)item - s_pref[r]);
            pix = on ? (int64_t)q.in_list[cas_at] : 0;
        } else if (listed) pix = on ? (int64_t)q.in_list[item] : 0;
        const unsigned boff = (unsigned)(on ? pix : 0) * 4u;     // byte offset inside a frame

        // A genuine +-Inf sample stays among the n valid ones, makes the variance
        // non-finite and thereby sends the pixel to the exact kernel (`bail`).
        // zonal sigma: only the clip zones and the median window need exact ranks (the
        // winsorized variant also reads single positions in between: full sort)
        constexpr int MW0 = ZONAL ? ZH / 2 - 1 : 0, MW1 = ZONAL ? ZL + NS / 2 + 1 : NS;
        using Sorter = std::conditional_t<(ZONAL && !WINSOR && NS >= 24), ZonalSort<ZL, MW0, MW1, ZH>, FullSort>;      // (16 positions: nothing to prune)
        float v[NS];
        // A stack with only one or two frames more than the next smaller network (17, 18, 25, 26, 33, 34, 49, 50 ...
        // frames) would keep a single sample or two in the high zone [ZH, NS) -- the rest of it is padding -- and hand every
        // pixel with a clipped high sample to the generic pass (8 ... 32 % of them, measured: 3 - 5 ms instead of 0.5 - 2).
        // Some of the padding therefore goes to the BOTTOM as -Inf, where it is what a clipped low sample is: dead
        // positions in front of the low pointer.  Both zones then hold at least about half their width in samples.
        int lo_pads = 0;
        // (winsorized kernels too -- 49 / 50 / 113 frames 23.7 / 11.1 / 22.7 -> 6.0 / 6.0 / 8.6 ms -- since the bound on the number of
        // clamped samples takes the dead positions in front of the low pointer into account, see n_lo below: without that
        // the padding widened every pixel's interval and 50 times as many went to the exact list)
        if constexpr (ZONAL && !TIGHT) {
            // (zones of 4 positions -- 17, 25, 33 frames: one frame above a network size -- split evenly, 2 + 2: with one
            // pad at the bottom the high zone held two samples and every pixel with two high clips went to the generic
            // pass -- 1.1 % of them at 25 frames, 187 k, and the appends alone cost the kernel 0.3 ms)
            lo_pads = min(max(KZ / 2 + 1 - (N - ZH), 0), KZ == 4 ? KZ / 2 : KZ / 2 - 1);     // (N is wave-uniform)
        }
        const int n = gather_sorted<NS, 16, Sorter, true, !TIGHT>(p.frames, p.stride, N, boff, v, lo_pads);
        bool to_exact = false;

        float res = p.ref_loc;
        int c_lo = 0, c_hi = 0;
        int a = lo_pads, b = lo_pads + n;       // surviving samples = sorted positions [a, b)
        bool active = on && n > 0;
        bool to_generic = false;
        if constexpr (ZONAL) {
            // zonal passes need b > ZH (and a < ZL): lanes with more missing
            // samples are handed to the generic pass, the others carry on
            to_generic = active && !(b > ZH);
            active = active && !to_generic;
        }

        // Shift c = first-pass median (any value near the bulk works).  With
        // D = sum(x-c) and Q = sum((x-c)^2) over the survivors,
        //     mean = c + D/cnt,   var = Q/cnt - (mean-c)^2        (exact identities).
        // Positions [ZL,ZH) are never clipped in the zonal passes, so their
        // share of D and Q is computed once; an iteration only re-sums the zones.
        constexpr int W0 = ZONAL ? ZH / 2 - 1 : 0, W1 = ZONAL ? ZL + NS / 2 + 1 : NS;
        float c = pick<W0, W1>(v, a + ((b - a) >> 1));
        float d_mid = 0.0f, q_mid = 0.0f;
        if constexpr (ZONAL) {
            float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            static_chunks<0, (ZH - ZL) / 4, 4>([&](auto K) NL_INL {
                constexpr int k = ZL + 4 * decltype(K)::value;
                const float e0 = v[k] - c, e1 = v[k + 1] - c, e2 = v[k + 2] - c, e3 = v[k + 3] - c;
                d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
            });
            d_mid = (d0 + d1) + (d2 + d3);
            q_mid = (q0 + q1) + (q2 + q3);
        }

        // winsorization (stack.go:646-672) clamps at median -/+ 1.5 sigma: in the zonal
        // passes only sorted positions outside [WL, WH) are allowed to reach a clamp,
        // the inner half contributes these fixed sums
        // (the clamps sit at +-1.5 sigma: 6.7 % of a Gaussian column per side, 8.6 +- 2.8 samples of 128;
        // a pixel with more goes to the replay through shape_ok)
        constexpr int WL = ZONAL ? NL_WINSOR_WL(NS) : 0, WH = ZONAL ? NS - WL - KP : NS;
        float d_in = 0.0f, q_in = 0.0f;
        if constexpr (ZONAL && WINSOR) {
            static_assert(WL >= ZL && WH <= ZH && (WL - ZL) % 4 == 0 && (WH - WL) % 4 == 0, "winsor zones");
            float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            static_chunks<0, (WH - WL) / 4, 4>([&](auto K) NL_INL {
                constexpr int k = WL + 4 * decltype(K)::value;
                const float e0 = v[k] - c, e1 = v[k + 1] - c, e2 = v[k + 2] - c, e3 = v[k + 3] - c;
                d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
            });
            d_in = (d0 + d1) + (d2 + d3);
            q_in = (q0 + q1) + (q2 + q3);
        }

        if constexpr (CONT) {
            // the clip counts an earlier stage left: the survivors are sorted positions [a + c_lo, b - c_hi) (the shift c
            // and the fixed sums above are those of the first stage: same arithmetic, same result bits)
            const unsigned st = on ? q.in_state[cas_at] : 0u;
            c_lo = (int)(st & 0xffffu);
            c_hi = (int)(st >> 16);
            a += c_lo;
            b -= c_hi;
        }
        // clipping passes this wave may still run in this stage (wave-uniform)
        int passes_left = (CASCADE && q.pass_budget > 0) ? q.pass_budget : 0x7fffffff;
        bool defer = false;                    // CASCADE: not done within the stage's budget -> continuation list

        // max|x| over the survivors (only enters the reference-mean error term):
        // first pass from the two ends of the sorted column, afterwards from the
        // bounds every survivor passed
        float amax = fmaxf(fabsf(pick<0, ZONAL ? ZL : 1>(v, a)), fabsf(pick<ZONAL ? ZH : 0, NS>(v, b - 1)));

        int rnd = 0;                           // RECORD: clipping rounds decided so far
        if (ZONAL && lane == 0) NL_STAT(4, 1);
        // one clipping pass of every active lane (stack.go:401-431; the winsorized variants with their loop inside)
        auto one_pass = [&]() NL_INL {
            if constexpr (CASCADE) {
                if (passes_left <= 0) {                    // whoever needs another pass takes it in the next stage
                    defer = defer || active;
                    active = false;
                    return;
                }
                passes_left--;
            }
            if (ZONAL) { if (lane == 0) NL_STAT(2, 1); if (active) NL_STAT(3, 1); }
            // re-materialised per pass: otherwise the differences v[k] - c of every masked
            // position are hoisted out of the loop (one register each -- 128 in the generic pass;
            // the 24 of the zonal sigma pass are left alone)
            float cz = c;
            if constexpr (!ZONAL || WINSOR) asm volatile("" : "+v"(cz));
            const int cnt = b - a;
            const float fcnt = (float)cnt;
            float dz0 = 0.0f, dz1 = 0.0f, qz0 = 0.0f, qz1 = 0.0f;
            if constexpr (ZONAL) {
                static_range<0, ZL>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const float e = (k >= a) ? v[k] - cz : 0.0f;
                    dz0 += e;
                    qz0 = __builtin_fmaf(e, e, qz0);
                });
                static_range<ZH, NS>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const float e = (k < b) ? v[k] - cz : 0.0f;
                    dz1 += e;
                    qz1 = __builtin_fmaf(e, e, qz1);
                });
            } else {
                const int a1 = opaque(a);
                float dz2 = 0.0f, dz3 = 0.0f, qz2 = 0.0f, qz3 = 0.0f;
                static_chunks<0, NS / 4, 2>([&](auto K) NL_INL {
                    constexpr int k = 4 * decltype(K)::value;
                    const bool i0 = (unsigned)(k + 0 - a1) < (unsigned)cnt;
                    const bool i1 = (unsigned)(k + 1 - a1) < (unsigned)cnt;
                    const bool i2 = (unsigned)(k + 2 - a1) < (unsigned)cnt;
                    const bool i3 = (unsigned)(k + 3 - a1) < (unsigned)cnt;
                    const float e0 = i0 ? v[k + 0] - cz : 0.0f, e1 = i1 ? v[k + 1] - cz : 0.0f;
                    const float e2 = i2 ? v[k + 2] - cz : 0.0f, e3 = i3 ? v[k + 3] - cz : 0.0f;
                    dz0 += e0; dz1 += e1; dz2 += e2; dz3 += e3;
                    qz0 = __builtin_fmaf(e0, e0, qz0); qz1 = __builtin_fmaf(e1, e1, qz1);
                    qz2 = __builtin_fmaf(e2, e2, qz2); qz3 = __builtin_fmaf(e3, e3, qz3);
                });
                dz0 += dz2; dz1 += dz3; qz0 += qz2; qz1 += qz3;
            }
            const float dsum = d_mid + (dz0 + dz1);
            const float qsum = q_mid + (qz0 + qz1);
            // (the means through a hardware reciprocal of the sample count, <= 1 ulp, and the square roots in hardware, <= 1 ulp:
            // these are OUR moments and the ends of an interval, not the reference's arithmetic -- the margins below pay for it:
            // two IEEE divisions and two correctly rounded square roots were 45 of a clipping pass's 540 instructions)
#ifdef NL_IEEE_PASS
            const float inv_cnt = 1.0f / fcnt;
            const float delta = dsum / fcnt;             // mean~ - c
#else
            const float inv_cnt = __builtin_amdgcn_rcpf(fcnt);
            const float delta = dsum * inv_cnt;          // mean~ - c
#endif
            const float m = c + delta;
#ifdef NL_IEEE_PASS
            const float aa = qsum / fcnt;                // E[(x-c)^2]~
#else
            const float aa = qsum * inv_cnt;             // E[(x-c)^2]~
#endif
            const float bb = delta * delta;
            const float var = fmaxf(aa - bb, 0.0f);

            // ---- bracket the reference's stddev ----
            // ours: aa carries <= NS/4+8 roundings per term; bb = delta^2 with delta off by
            // <= (NS/4+7) u mean|e|, and 2|delta| mean|e| <= aa + bb: together <= (NS/2+17) u (aa+bb)
#ifdef NL_IEEE_PASS
            const float err_o = ((float)(NS / 2 + 24)) * kU * (aa + bb);
#else
            const float err_o = ((float)(NS / 2 + 32)) * kU * (aa + bb);     // (+8: the reciprocal's ulp in delta, delta^2 and aa)
#endif
            // reference: relative gamma_(n+3) on its variance, its mean off by <= e_m
            const float eps_r = 1.02f * (fcnt + 8.0f) * kU;
            const float e_m = 1.02f * (fcnt + 2.0f) * kU * amax;
            const float v_up = var + err_o;
            const float v_dn = fmaxf(var - err_o, 0.0f);
            const float v_hi = v_up + v_up * eps_r + e_m * e_m;
            const float v_lo = fmaxf(v_dn - v_dn * eps_r, 0.0f);
#ifdef NL_IEEE_PASS
            float s_max = __fsqrt_rn(v_hi) * (1.0f + 4.0f * kU);
            float s_min = __fsqrt_rn(v_lo) * (1.0f - 4.0f * kU);
#else
            // (hardware square root: <= 1 ulp = 2u, flushes denormals -- the absolute term; a flushed lower end is 0)
            float s_max = __builtin_amdgcn_sqrtf(v_hi) * (1.0f + 6.0f * kU) + 4.0e-19f;
            float s_min = __builtin_amdgcn_sqrtf(v_lo) * (1.0f - 6.0f * kU);
#endif
            bool bail = !(v_hi < 3.0e38f);          // overflow / NaN (e.g. an Inf sample): exact kernel

            // ---- exact median (qsort.go:68-82): sorted column, position lookup ----
            // zonal: a in [0,ZL), b in (ZH,NS]  =>  kk in [ZH/2, ZL-1+NS/2]
            const int kk = a + (cnt >> 1);
            float upper, lower;
            pick_pair<W0, W1>(v, kk, lower, upper);
            const float median = (cnt & 1) ? upper : 0.5f * (lower + upper);

            // ---- the reference's bound expressions at both ends of an interval [smin, smax] of its stddev (stack.go:408-409;
            // fp32 multiply then add, never fused), and the certain (c1, d1) / possible (c2, d2) clips they imply.  The
            // column is sorted, so the samples below a threshold are a prefix and those above it a suffix (pads are
            // +Inf): counted over the whole zone without rank masks, minus what is already excluded ----
            float lo_min, lo_max, hi_min, hi_max;
            auto clip_counts = [&](const float smin, const float smax, int &c1, int &c2, int &d1, int &d2) NL_INL {
                const float tl0 = __fmul_rn(p.sig_lo, smin), tl1 = __fmul_rn(p.sig_lo, smax);
                const float th0 = __fmul_rn(p.sig_hi, smin), th1 = __fmul_rn(p.sig_hi, smax);
                const float la = __fsub_rn(median, tl0), lb = __fsub_rn(median, tl1);
                const float ha = __fadd_rn(median, th0), hb = __fadd_rn(median, th1);
                lo_min = fminf(la, lb); lo_max = fmaxf(la, lb);
                hi_min = fminf(ha, hb); hi_max = fmaxf(ha, hb);
                c1 = 0; c2 = 0; d1 = 0; d2 = 0;
                if constexpr (ZONAL) {
                    static_range<0, ZL>([&](auto K) NL_INL {
                        constexpr int k = decltype(K)::value;
                        c1 += (v[k] < lo_min) ? 1 : 0;
                        c2 += (v[k] < lo_max) ? 1 : 0;
                    });
                    static_range<ZH, NS>([&](auto K) NL_INL {
                        constexpr int k = decltype(K)::value;
                        d1 += (v[k] > hi_max) ? 1 : 0;
                        d2 += (v[k] > hi_min) ? 1 : 0;
                    });
                    c1 = max(c1 - a, 0); c2 = max(c2 - a, 0);
                    d1 = max(d1 - (NS - b), 0); d2 = max(d2 - (NS - b), 0);
                } else {
                    static_chunks<0, NS, 8>([&](auto K) NL_INL {
                        constexpr int k = decltype(K)::value;
                        const float x = v[k];
                        c1 += (x < lo_min) ? 1 : 0;
                        c2 += (x < lo_max) ? 1 : 0;
                        d1 += (x > hi_max) ? 1 : 0;
                        d2 += (x > hi_min) ? 1 : 0;
                    });
                    c1 = min(max(c1 - a, 0), cnt); c2 = min(max(c2 - a, 0), cnt);
                    d1 = min(max(d1 - (NS - b), 0), cnt); d2 = min(max(d2 - (NS - b), 0), cnt);
                }
            };

            if constexpr (WINSOR) {
                // ---- winsorized stddev, stack.go:646-672, as an interval ----
                // The reference repeats { clamp a copy to median -/+ 1.5*std; std =
                // 1.134*stddev(copy) } until nothing changed or std moved by <= 0.05 %.
                // Its std is again an order-dependent fp32 sum, so we carry an interval
                // [w_lo, w_hi] for it through the loop (WinsorInterval, fast_common.hpp).
                constexpr int PZ = ZONAL ? ZH : 0;
                const float xmin = pick<0, ZONAL ? ZL : NS>(v, a);
                const float xmax = pick<PZ, NS>(v, b - 1);
                WinsorInterval wi;
                // (generic pass: a wave runs for its slowest pixel -- the few pixels with very long loops are cheaper in the replay)
                // (and the stages of the cascade behind the dominant kernel -- the last one has no budget of its own)
                wi.start(s_min, s_max, ((!ZONAL || CONT) && q.gen_round_cap > 0) ? q.gen_round_cap : 100);
                bool inner = active && !bail;
                int rounds_left = (CASCADE && q.round_cap > 0) ? q.round_cap : 0x7fffffff;
                // (round 5) INVARIANT-INTERVAL CERTIFICATE.  The loop is monotone: the clamps only tighten, and the variance
                // of the clamped copy is monotone in the clamp.  Let L <= (every possible value of) the current std, and let
                // the lower end of the NEXT std's interval, evaluated at the clamp L itself would produce -- tighter than any
                // clamp a std >= L can produce, composed with the clamps so far -- be >= L.  Then by induction every later
                // std of the reference is >= L, and w_hi bounds them from above (clamps at least as tight as today's
                // loosest): wherever the reference leaves its loop, its std lies in [L, w_hi].  If the clip decisions agree
                // over that interval (and over what the hull already holds) the loop is left NOW.  L is an Aitken
                // extrapolation of the last three lower ends with a margin; a failed trial costs one evaluation.
                // tools/winsor_cert_sim.py: rounds per lock-step wave 46 -> 27 at 16 frames, 38 -> 24 at 24, 25 -> 20 at 128.
                // A trial is an iteration of the SAME loop -- the one evaluation of the clamped variance, at the trial's clamp
                // instead of the round's: a second inlined copy of that evaluation took the 128-position kernel from 165 to 237
                // registers (two waves per SIMD instead of three: 4.5 -> 5.7 ms).
                constexpr bool CERT = NL_CERT_ON(NS) && (ZONAL || NS <= 32) && !RECORD;
                float wl_h0 = s_min, wl_h1 = s_min;        // lower ends of the std one and two rounds ago
                int wr = 0;
                bool trial_next = false;                   // wave-uniform
                while (__any(inner)) {
                    const bool is_trial = CERT && trial_next;
                    trial_next = false;
                    float Lc, Hc, ltry = 0.0f;
                    bool trial = false;
                    if (!is_trial) {
                    if constexpr (CASCADE) {
                        if (rounds_left <= 0) {            // at the cap: these lanes re-enter this pass in the next stage
                            defer = defer || inner;
                            active = active && !inner;
                            break;
                        }
                        rounds_left--;
                    }
                    if (ZONAL) { if (lane == 0) NL_STAT(0, 1); if (inner) NL_STAT(1, 1); }
                    wl_h0 = wl_h1; wl_h1 = wi.w_lo;
                    wi.next_clamp(median, xmin, xmax);
                    Lc = wi.Lp; Hc = wi.Hm;
                    } else {
                        const float s2 = wi.w_lo, s1 = wl_h1, s0 = wl_h0;
                        const float dd1 = s1 - s2, dd0 = s0 - s1;
                        float rr = dd1 * __builtin_amdgcn_rcpf(dd0);
                        rr = fminf(fmaxf(rr, 0.0f), 0.95f);
                        const float rest = dd1 * rr * __builtin_amdgcn_rcpf(1.0f - rr);
                        ltry = fmaxf(s2 - 1.5f * rest - 1.0e-4f * s2, 0.0f);
                        trial = inner && dd1 > 0.0f && dd0 > 0.0f && ltry > 0.0f && ltry <= s2;
                        // the clamp the value ltry itself would produce (the reference's own operations: monotone in the std)
                        const float tq = __fmul_rn(1.5f, ltry);
                        Lc = fmaxf(wi.Lp, __fsub_rn(median, tq));
                        Hc = fminf(wi.Hm, __fadd_rn(median, tq));
                    }
                    // variance of clamp(x, Lt, Ht) over the survivors (shifted moments) and its error bound
                    auto clamped_variance = [&](const float Lt, const float Ht, float &wvar, float &werr, float &wmean_c,
                                                float &wrms) NL_INL {
                    float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                    if constexpr (ZONAL) {
                        // only the outer quarters of the sorted column can sit on a clamp
                        // (checked below); the inner half enters unclamped through d_in / q_in
                        static_range<0, ZL>([&](auto K) NL_INL {
                            constexpr int k = decltype(K)::value;
                            const float e = (k >= a) ? max_raw(v[k], Lt) - cz : 0.0f;
                            d0 += e; q0 = __builtin_fmaf(e, e, q0);
                        });
                        static_chunks<0, (WL - ZL) / 4, 2>([&](auto K) NL_INL {
                            constexpr int k = ZL + 4 * decltype(K)::value;
                            const float e0 = max_raw(v[k], Lt) - cz, e1 = max_raw(v[k + 1], Lt) - cz;
                            const float e2 = max_raw(v[k + 2], Lt) - cz, e3 = max_raw(v[k + 3], Lt) - cz;
                            d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                            q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                            q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
                        });
                        static_chunks<0, (ZH - WH) / 4, 2>([&](auto K) NL_INL {
                            constexpr int k = WH + 4 * decltype(K)::value;
                            const float e0 = min_raw(v[k], Ht) - cz, e1 = min_raw(v[k + 1], Ht) - cz;
                            const float e2 = min_raw(v[k + 2], Ht) - cz, e3 = min_raw(v[k + 3], Ht) - cz;
                            d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                            q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                            q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
                        });
                        static_range<ZH, NS>([&](auto K) NL_INL {
                            constexpr int k = decltype(K)::value;
                            const float e = (k < b) ? min_raw(v[k], Ht) - cz : 0.0f;
                            d1 += e; q1 = __builtin_fmaf(e, e, q1);
                        });
                        d2 += d_in; q2 += q_in;
                    } else {
                        const int a4 = opaque(a);
                        static_chunks<0, NS / 4, 2>([&](auto K) NL_INL {
                            constexpr int k = 4 * decltype(K)::value;
                            const bool i0 = (unsigned)(k + 0 - a4) < (unsigned)cnt;
                            const bool i1 = (unsigned)(k + 1 - a4) < (unsigned)cnt;
                            const bool i2 = (unsigned)(k + 2 - a4) < (unsigned)cnt;
                            const bool i3 = (unsigned)(k + 3 - a4) < (unsigned)cnt;
                            const float e0 = i0 ? __builtin_amdgcn_fmed3f(v[k + 0], Lt, Ht) - cz : 0.0f;
                            const float e1 = i1 ? __builtin_amdgcn_fmed3f(v[k + 1], Lt, Ht) - cz : 0.0f;
                            const float e2 = i2 ? __builtin_amdgcn_fmed3f(v[k + 2], Lt, Ht) - cz : 0.0f;
                            const float e3 = i3 ? __builtin_amdgcn_fmed3f(v[k + 3], Lt, Ht) - cz : 0.0f;
                            d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                            q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                            q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
                        });
                    }
                    const float wd = ((d0 + d1) + (d2 + d3)) * inv_cnt;      // reciprocal: 2 more roundings,
                    const float wa = ((q0 + q1) + (q2 + q3)) * inv_cnt;      // covered by werr
                    const float wb = wd * wd;
                    wvar = fmaxf(wa - wb, 0.0f);
#ifdef NL_IEEE_PASS
                    werr = ((float)(NS / 2 + 34)) * kU * (wa + wb);
#else
                    werr = ((float)(NS / 2 + 40)) * kU * (wa + wb);
#endif
                    wmean_c = wd;                                  // mean of the copy, minus c
                    wrms = wa;                                     // E[(copy - c)^2]
                    };
                    float var_t, err_t, wd_t, wa_t;
                    clamped_variance(Lc, Hc, var_t, err_t, wd_t, wa_t);
                    if (is_trial) {
                        const float x_dn = fmaxf(var_t - err_t, 0.0f);
                        const float x_lo = __builtin_amdgcn_sqrtf(fmaxf(x_dn - x_dn * eps_r, 0.0f)) * (1.0f - 4.0f * kU);
                        trial = trial && __fmul_rn(1.134f, x_lo) >= ltry && (!ZONAL || (v[WL] >= Lc && v[WH - 1] <= Hc));
                        // every value the reference may leave the loop with: what the hull holds, and [ltry, w_hi]
                        const float h_lo = fminf(wi.hull_lo, ltry), h_hi = fmaxf(wi.hull_hi, wi.w_hi);
                        int e1, e2, f1, f2;
                        clip_counts(h_lo, h_hi, e1, e2, f1, f2);
                        trial = trial && e1 == e2 && f1 == f2 && h_hi < 3.0e38f;
                        if (trial) {
                            wi.hull_lo = h_lo; wi.hull_hi = h_hi;
                            inner = false;
                        }
                        continue;
                    }
                    // The loosest clamp (Lm, Hp) is not evaluated: with y = the copy at the tightest
                    // clamp and z = the copy at the loosest, z - y = delta is non-zero only for the
                    // n_lo samples below Lp (delta in [-(Lp-Lm), 0], y = Lp there) and the n_hi samples
                    // above Hm (delta in [0, Hp-Hm], y = Hm), so
                    //   var(z) = var(y) + 2 cov(y, delta) + var(delta)
                    //         <= var(y) + [n_lo dL (2 (ybar-Lp) + dL) + n_hi dH (2 (Hm-ybar) + dH)] / cnt.
                    // n_lo, n_hi only need upper bounds: the column is sorted, so testing every 4th
                    // position (every 2nd, every one for the small networks) bounds them to +3.
                    // (CS = 1 for the small networks: +3 on a handful of clamped samples would loosen the bound)
                    constexpr int CS = NS >= 64 ? 4 : (NS >= 48 ? 2 : 1);
                    int t_lo = 0, t_hi = 0;
                    if constexpr (ZONAL) {
                        static_range<0, WL / CS>([&](auto J) NL_INL {
                            constexpr int k = CS * decltype(J)::value + CS - 1;
                            const bool below = v[k] < wi.Lp;
                            t_lo += ((k >= ZL || k >= a) && below) ? 1 : 0;
                        });
                        static_range<0, (NS - WH) / CS>([&](auto J) NL_INL {
                            constexpr int k = WH + CS * decltype(J)::value;
                            const bool above = v[k] > wi.Hm;
                            t_hi += ((k < ZH || k < b) && above) ? 1 : 0;
                        });
                    } else {
                        const int a5 = opaque(a);
                        static_range<0, NS / CS>([&](auto J) NL_INL {
                            constexpr int k = CS * decltype(J)::value;
                            const bool in_lo = (unsigned)(k + CS - 1 - a5) < (unsigned)cnt;
                            const bool in_hi = (unsigned)(k - a5) < (unsigned)cnt;
                            t_lo += (in_lo && v[k + CS - 1] < wi.Lp) ? 1 : 0;
                            t_hi += (in_hi && v[k] > wi.Hm) ? 1 : 0;
                        });
                    }
                    float var_l, err_l;
                    {
                        // (the tested positions are k = CS - 1 (mod CS) from below and k = 0 (mod CS) from above, the survivors
                        // are [a, b): t tested survivors beyond a clamp mean at most CS t + CS - 1 - a % CS, resp.
                        // CS t + (b - 1) % CS survivors beyond it -- without the remainders every dead position in front
                        // of the low pointer, clipped or -Inf padding, widened the interval)
                        const float n_lo = (float)min(CS * t_lo + CS - 1 - (a % CS), cnt);
                        const float n_hi = (float)min(CS * t_hi + ((b - 1) % CS), cnt);
                        const float dL = (wi.Lp - wi.Lm) * (1.0f + 2.0f * kU), dH = (wi.Hp - wi.Hm) * (1.0f + 2.0f * kU);
                        // ybar = cz + wd_t, off by <= (NS/4+8) u mean|y-c| <= 3e-6 sqrt(E[(y-c)^2]) plus its own rounding
                        const float ybar = cz + wd_t;
                        const float slop = 4.0e-6f * __builtin_amdgcn_sqrtf(wa_t) + 4.0f * kU * fabsf(ybar) + 1.0e-30f;
                        const float gL = fmaxf(ybar - wi.Lp, 0.0f) + slop, gH = fmaxf(wi.Hm - ybar, 0.0f) + slop;
                        const float corr = (n_lo * (dL * (2.0f * gL + dL)) + n_hi * (dH * (2.0f * gH + dH))) * inv_cnt;
                        // an infinite clamp width (first round: Lm = Lp = -Inf gives Inf - Inf) cannot occur:
                        // both ends of an interval are finite or the same infinity -> NaN -> 0
                        var_l = var_t + ((corr == corr) ? corr * 1.001f : 0.0f);
                        err_l = err_t;
                    }
                    // zonal: the inner half must be strictly inside every clamp of the interval
                    const bool shape_ok = !ZONAL || (v[WL] >= wi.Lp && v[WH - 1] <= wi.Hm);
                    wi.finish_round(var_t, err_t, var_l, err_l, eps_r, e_m, shape_ok, inner, bail);
                    wr++;
                    if constexpr (CERT) trial_next = q.cert_first > 0 && wr >= q.cert_first && ((wr - q.cert_first) % q.cert_every) == 0;
                }
                s_min = wi.hull_lo;
                s_max = wi.hull_hi;
            }

            // ---- count certain clips (c1,d1) and possible clips (c2,d2) over the interval of the stddev ----
            int c1, c2, d1, d2;
            clip_counts(s_min, s_max, c1, c2, d1, d2);
            if constexpr (ZONAL) {
                // the zones must still hold a survivor on each side, otherwise the
                // next sorted position (outside the zone) might be clipped as well:
                // such a lane restarts in the generic pass
                if (active && ((a + c2 >= ZL) || (b - d2 <= ZH))) {
                    to_generic = true;
                    active = false;
                }
            }
            if (active) {
                // a sample inside the window, or (negative sigma) inverted bounds where the
                // reference's "low first" order matters: let the exact kernel decide
                bail |= (c1 != c2) || (d1 != d2) || (lo_max > hi_min && (c1 + d1) > 0);
                if (bail) {
                    to_exact = true;
                    active = false;
                    // winsorized passes: the rounds decided so far are on record for the replay (see below)
                    // (the generic instantiation records nothing: 0 rounds)
                    if constexpr (WINSOR && !RECORD) { if (p.nrounds) p.nrounds[pix] = (unsigned char)(ZONAL ? min(rnd, kBoundRounds) : 0); }
                } else {
                    if constexpr (WINSOR && ZONAL && !RECORD) {
                        // A pixel that turns undecidable in a LATER round is replayed from scratch -- but the
                        // rounds decided before that need no winsorization loop in the replay (about 20 sequential
                        // sums each): their thresholds go on record as in the decision pass of weighted stacks
                        if (p.bounds) {
                            if (rnd < kBoundRounds) p.bounds[(size_t)rnd * (size_t)p.npix + (size_t)pix] = make_float2(lo_max, hi_min);
                            rnd++;
                        }
                    }
                    if constexpr (RECORD) {
                        // x < lo_max <=> x < the reference's bound, x > hi_min <=> x > its bound, for every surviving
                        // sample: none lies in [lo_min, lo_max) or (hi_min, hi_max] (c1 == c2, d1 == d2)
                        if (rnd < kBoundRounds) p.bounds[(size_t)rnd * (size_t)p.npix + (size_t)pix] = make_float2(lo_max, hi_min);
                        rnd++;
                    }
                    c_lo += c1;
                    c_hi += d1;
                    a += c1;
                    b -= d1;
                    amax = fminf(amax, fmaxf(fabsf(lo_min), fabsf(hi_max)));   // survivors lie in [lo_min, hi_max]
                    if ((c1 + d1) == 0 || (b - a) <= 1) {     // stack.go:427-430: mean BEFORE this pass
                        res = m;
                        active = false;
                    }
                }
            }
        };
        // what a lane that is through leaves behind: its result and counts, or its place on a hand-over / continuation list
        auto flush = [&]() NL_INL {
        if (on && !active && !to_generic && !to_exact && !defer) {
            NL_STORE_RESULT(&p.out[pix], res);
            c_lo_total += c_lo;
            c_hi_total += c_hi;
        }
        if constexpr (CASCADE) {
            const unsigned long long dm = __ballot(on && defer);
            if (dm) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_cont, (unsigned)__popcll(dm));       // LDS
                base = __shfl(base, 0, 64);
                const unsigned slot = base + (unsigned)__popcll(dm & ((1ull << lane) - 1ull));
                if (on && defer && slot < q.cont_region) {
                    const size_t at = (size_t)blockIdx.x * q.cont_region + slot;
                    q.cont_list[at] = (unsigned)pix;
                    q.cont_state[at] = (unsigned)c_lo | ((unsigned)c_hi << 16);
                }
            }
        }
        // hand-over lists: a contiguous run per WORKGROUP -- the waves take their places in it through an LDS counter,
        // one device atomic reserves the run (round 4: one atomic per wave cost 11 ns each, all on one L2 channel -- a
        // frame count just above a network size sends 1 - 3 % of the pixels here from half of the 262 144 waves: 1.4 ms
        // of serialised atomics inside a 0.4 ms kernel at 25 frames), lanes fill it in order, so the consumer's loads
        // stay coalesced
        if constexpr (ZONAL && !RECORD) {
            const unsigned long long gm = __ballot(on && to_generic);
            unsigned woff = 0;
            if (lane == 0 && gm) woff = atomicAdd(&s_gen, (unsigned)__popcll(gm));          // LDS
            __syncthreads();
            if (threadIdx.x == 0) s_gen_base = s_gen ? atomicAdd(q.gen_count, s_gen) : 0u;
            __syncthreads();
            if (gm) {
                const unsigned base = s_gen_base + (unsigned)__shfl((int)woff, 0, 64);
                const unsigned slot = base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull));
                if (on && to_generic && slot < q.gen_capacity) q.gen_list[slot] = (unsigned)pix;
            }
            __syncthreads();
            if (threadIdx.x == 0) s_gen = 0u;                  // (for the next trip of a listed kernel; ordered by the barriers above / below)
        }
        const unsigned long long em = RECORD ? 0ull : __ballot(on && to_exact);
        if (em) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
            base = __shfl(base, 0, 64);
            const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
            if (on && to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
        }
            on = on && active;                         // (whoever is not active any more is done with)
            to_generic = false;
            to_exact = false;
            defer = false;
        };
        // (Round 4 also tried an intra-workgroup compaction here: after two clipping passes the unfinished lanes of the
        // four waves moved what a pass touches -- clip / clamp zones, median window, a handful of scalars -- through LDS
        // into one or two packed waves.  A wave runs 4.5 passes where a lane needs 2.0 (sigma, 128 frames; winsorized:
        // 3.4 passes and 28 winsorization rounds against 1.8 and 9.6), yet plain sigma clipping LOST 6 % at 100 and 128
        // frames (headline kernel 1.55 - 1.59 -> 1.67 ms; a pass is only 430 instructions) and the winsorized kernels
        // gained nothing (4.59 -> 4.60 ms).  Removed; profiles/r04_sigma512_experiments.txt.)
        while (__any(active)) one_pass();
        if constexpr (RECORD) {
            // (a pixel without data is left to the full replay, which writes RefFrameLoc)
            const bool decided = on && !to_generic && !to_exact && n > 0 && rnd <= kBoundRounds;
            if (on) p.nrounds[pix] = (unsigned char)(decided ? rnd : 0);
        } else {
            flush();
        }
    }
    if constexpr (RECORD) return;
    if constexpr (CASCADE) {
        if (q.cont_count) {                                    // this workgroup's region length (every workgroup stores one: no zeroing)
            __syncthreads();
            if (threadIdx.x == 0) q.cont_count[blockIdx.x] = min(s_cont, q.cont_region);
        }
    }

    // clip totals: wave sum -> block sum -> one slot per workgroup
    __shared__ int s_lo[4], s_hi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_lo_total += __shfl_xor(c_lo_total, o, 64);
        c_hi_total += __shfl_xor(c_hi_total, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = c_lo_total; s_hi[threadIdx.x >> 6] = c_hi_total; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t_lo = 0, t_hi = 0;
        for (unsigned w = 0; w < (blockDim.x >> 6); w++) { t_lo += s_lo[w]; t_hi += s_hi[w]; }
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if constexpr (!ZONAL) slot = clip_slot(p, blockIdx.x);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}


}  // namespace nl
