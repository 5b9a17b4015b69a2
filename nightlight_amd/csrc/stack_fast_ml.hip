// stack_fast_ml.hip -- register-resident sigma clipping for 129..512 frames:
// LPP = 2 or 4 ADJACENT lanes share one pixel, each lane keeps 128 samples in
// VGPRs (so the register footprint equals the 128-frame kernel's).
//
//   * lane role r loads frames r, r+LPP, r+2*LPP, ... of the pixel;
//   * each lane sorts its 128 values with the odd-even merge network;
//   * the 2 (4) sorted runs are merged across lanes with bitonic merge stages:
//     the cross-lane compare-exchanges read the partner's register through a
//     DPP quad permute (no LDS), the remaining stages are in-lane;
//     afterwards lane r holds the pixel's sorted ranks [128 r, 128 r + 128);
//   * everything after the sort is the algorithm of stack_fast.hip (exact
//     median by rank lookup, shifted moments, rigorous bracket of the
//     reference's stddev, clip decisions accepted only when unambiguous, exact
//     kernel for the rest -- see that file and DESIGN.md section 5); sums,
//     counts and lookups are combined over the quad with DPP adds / ors, which
//     are commutative, so all lanes of a pixel hold bit-identical values and
//     take the same branches.
//
// StackSigma: internal/ops/stack/stack.go:372-436.  HBM traffic: every sample is
// read once; a wave instruction covers 64/LPP consecutive pixels of LPP frames.
#include <cstdlib>
#include <string>

#include "fast_ml_common.hpp"

namespace nl {

// WIDE (zonal only): for frame counts well below LPP*128.  The unused positions sort
// to the top as +Inf, so the last lanes hold nothing but padding and the high
// zone has to reach down to the last real samples: it covers a whole lane
// except its 8 lowest ranks, the last lane holding data (LAST) is found at run
// time from the frame count, and the median is looked up over whole lanes.
// Valid while more than LAST*128 + 8 samples are present.
// (zonal sigma: the allocator lands one register above the 168 that let 3 waves share a SIMD)
#ifdef NL_ROUND_STATS
__device__ unsigned long long nl_dbg_rounds_ml[8];           // as nl_dbg_rounds in stack_fast.hip
extern "C" int nl_debug_round_stats_ml(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nl_dbg_rounds_ml), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nl_dbg_rounds_ml), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define NL_STAT(i, x) atomicAdd(&nl_dbg_rounds_ml[i], (unsigned long long)(x))
#else
#define NL_STAT(i, x) ((void)0)
#endif

#ifdef NL_EXPERIMENTS      // the round-1 kernel (zones in registers): superseded by the LDS-column kernels of stack_fast_mlz*.hip /
                           // stack_fast_mlg.hip for every frame count 129 ... 512; kept for A/B runs (NL_MLZ=0, NL_MLG=0) in the experiments build
template <int LPP, bool ZONAL, bool WINSOR, bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ZONAL ? (WINSOR ? 2 : 3) : 1, 8)))
void stack_sigma_ml_kernel(StackArgs p, FastArgs q)
{
    if constexpr (!ZONAL) { if (q.in_list) snapshot_fb_list(q); }
    constexpr int NS = kMlNS, NT = NS * LPP;
    constexpr int ZL = kZone;                        // low zone : ranks [0, ZL)           (role 0)
    static_assert(!WIDE || ZONAL, "WIDE is a zonal variant");
    constexpr int ZHS = WIDE ? NS - kZone : kZone + kPadMax;   // high zone: the ZHS highest ranks of lane LAST

    int c_lo_total = 0, c_hi_total = 0;
    const bool listed = !ZONAL && q.in_list != nullptr;
    const int64_t limit = listed ? (int64_t)min(*q.in_count, q.in_capacity) : p.npix;
    const int64_t items_per_wg = blockDim.x / LPP;
    const int64_t sweep = listed ? (int64_t)gridDim.x * items_per_wg : limit;
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x % LPP;

    for (int64_t wg_item = (int64_t)blockIdx.x * items_per_wg; wg_item < limit; wg_item += sweep) {
        int N = p.n_frames;
        asm volatile("" : "+s"(N));
        // last lane that holds samples, ranks in use, first rank of the high zone
        const int LAST = WIDE ? (N - 1) / NS : LPP - 1;
        const int NTE = (LAST + 1) * NS;
        const int ZH = ZONAL ? NTE - ZHS : NT;
        const int64_t item = wg_item + threadIdx.x / LPP;
        const bool on = item < limit;
        int64_t pix = item;
        if (listed) pix = on ? (int64_t)q.in_list[item] : 0;

        float v[NS];
        // (winsorized: every lane fully sorted -- the clamped counts below sample every 8th rank)
        const int n = ml_gather_sorted<LPP, NS, ZONAL && !WIDE && !WINSOR>(p.frames, p.stride, N, on, pix, role, v);

        bool to_exact = false;
        float res = p.ref_loc;
        int c_lo = 0, c_hi = 0;
        int a = 0, b = n;                       // survivors = global sorted ranks [a, b)
        bool active = on && n > 0;
        bool to_generic = false;
        if constexpr (ZONAL) {
            to_generic = active && !(n > ZH);
            active = active && !to_generic;
        }

        // median windows of the zonal passes: a in [0,ZL), b in (ZH,NT] => the two
        // middle ranks lie in [NT/2 - ZHS/2 - 1, NT/2 + ZL/2], i.e. at the top of
        // lane LPP/2-1 and at the bottom of lane LPP/2
        constexpr int TOPW = (ZONAL && !WIDE) ? ZHS / 2 + 2 : NS, BOTW = (ZONAL && !WIDE) ? ZL / 2 + 2 : NS;
        constexpr int MIDR = LPP / 2 - 1;
        const float c = pick_rank<LPP, NS, TOPW, BOTW>(v, a + ((b - a) >> 1), role, MIDR);

        // shifted moments of the never-clipped ranks [ZL, ZH), once
        float d_mid = 0.0f, q_mid = 0.0f;
        if constexpr (ZONAL) {
            float d0 = 0, d1 = 0, q0 = 0, q1 = 0;          // [0,ZL), [NS-ZHS,NS) of this lane
            static_range<0, ZL>([&](auto K) NL_INL {
                constexpr int k = decltype(K)::value;
                const float e = v[k] - c;
                d0 += e; q0 = __builtin_fmaf(e, e, q0);
            });
            static_range<NS - ZHS, NS>([&](auto K) NL_INL {
                constexpr int k = decltype(K)::value;
                const float e = v[k] - c;
                d1 += e; q1 = __builtin_fmaf(e, e, q1);
            });
            float m0 = 0, m1 = 0, m2 = 0, m3 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            static_chunks<0, (NS - ZHS - ZL) / 4, 4>([&](auto K) NL_INL {
                constexpr int k = ZL + 4 * decltype(K)::value;
                const float e0 = v[k] - c, e1 = v[k + 1] - c, e2 = v[k + 2] - c, e3 = v[k + 3] - c;
                m0 += e0; m1 += e1; m2 += e2; m3 += e3;
                r0 = __builtin_fmaf(e0, e0, r0); r1 = __builtin_fmaf(e1, e1, r1);
                r2 = __builtin_fmaf(e2, e2, r2); r3 = __builtin_fmaf(e3, e3, r3);
            });
            // the low zone of lane 0 and the high zone of the last lane are re-summed per pass
            float dl = (m0 + m1) + (m2 + m3), ql = (r0 + r1) + (r2 + r3);
            dl += (role == 0 ? 0.0f : d0) + (role == LAST ? 0.0f : d1);
            ql += (role == 0 ? 0.0f : q0) + (role == LAST ? 0.0f : q1);
            if constexpr (WIDE) {                  // lanes above LAST hold only padding
                dl = role > LAST ? 0.0f : dl;
                ql = role > LAST ? 0.0f : ql;
            }
            d_mid = quad_sum<LPP>(dl);
            q_mid = quad_sum<LPP>(ql);
        }

        float amax;
        {
            const float lowest = __int_as_float(quad_or<LPP>(role == 0 ? __float_as_int(v[0]) : 0));
            const float highest = pick_rank<LPP, NS, ZONAL ? ZHS : NS, 1>(v, n - 1, role, ZONAL ? LAST : -2);
            amax = fmaxf(fabsf(lowest), fabsf(highest));
        }

        if (ZONAL && lane == 0) NL_STAT(4, 1);
        while (__any(active)) {
            if (ZONAL) { if (lane == 0) NL_STAT(2, 1); if (active && role == 0) NL_STAT(3, 1); }
            // WIDE: re-materialised per pass, otherwise the 120 differences v[k] - c of the
            // wide zone are hoisted out of the loop and cost 120 registers
            float cz = c;
            if constexpr (WIDE || !ZONAL) asm volatile("" : "+v"(cz));
            const int cnt = b - a;
            const float fcnt = (float)cnt;
            float dz = 0.0f, qz = 0.0f;
            if constexpr (ZONAL) {
                float dz0 = 0, qz0 = 0, dz1 = 0, qz1 = 0;
                static_range<0, ZL>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const float e = (k >= a) ? v[k] - cz : 0.0f;
                    dz0 += e;
                    qz0 = __builtin_fmaf(e, e, qz0);
                });
                const int b_local = b - LAST * NS;
                static_range<NS - ZHS, NS>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const float e = (k < b_local) ? v[k] - cz : 0.0f;
                    dz1 += e;
                    qz1 = __builtin_fmaf(e, e, qz1);
                });
                dz = quad_sum<LPP>((role == 0 ? dz0 : 0.0f) + (role == LAST ? dz1 : 0.0f));
                qz = quad_sum<LPP>((role == 0 ? qz0 : 0.0f) + (role == LAST ? qz1 : 0.0f));
            } else {
                const int a1 = opaque(a - role * NS);
                float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                static_chunks<0, NS / 4, 2>([&](auto K) NL_INL {
                    constexpr int k = 4 * decltype(K)::value;
                    const bool i0 = (unsigned)(k + 0 - a1) < (unsigned)cnt;
                    const bool i1 = (unsigned)(k + 1 - a1) < (unsigned)cnt;
                    const bool i2 = (unsigned)(k + 2 - a1) < (unsigned)cnt;
                    const bool i3 = (unsigned)(k + 3 - a1) < (unsigned)cnt;
                    const float e0 = i0 ? v[k + 0] - cz : 0.0f, e1 = i1 ? v[k + 1] - cz : 0.0f;
                    const float e2 = i2 ? v[k + 2] - cz : 0.0f, e3 = i3 ? v[k + 3] - cz : 0.0f;
                    d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                    q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                    q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
                });
                dz = quad_sum<LPP>((d0 + d1) + (d2 + d3));
                qz = quad_sum<LPP>((q0 + q1) + (q2 + q3));
            }
            const float dsum = d_mid + dz;
            const float qsum = q_mid + qz;
            const float delta = dsum / fcnt;             // mean~ - c
            const float m = c + delta;
            const float aa = qsum / fcnt;                // E[(x-c)^2]~
            const float bb = delta * delta;
            const float var = fmaxf(aa - bb, 0.0f);

            // ---- bracket the reference's stddev ----
            // ours: see stack_fast.hip (NS/2 + 17), plus the log-depth quad adds and a margin
            const float err_o = ((float)(NS / 2 + 40)) * kU * (aa + bb);
            const float eps_r = 1.02f * (fcnt + 8.0f) * kU;
            const float e_m = 1.02f * (fcnt + 2.0f) * kU * amax;
            const float v_up = var + err_o;
            const float v_dn = fmaxf(var - err_o, 0.0f);
            const float v_hi = v_up + v_up * eps_r + e_m * e_m;
            const float v_lo = fmaxf(v_dn - v_dn * eps_r, 0.0f);
            float s_max = __fsqrt_rn(v_hi) * (1.0f + 4.0f * kU);
            float s_min = __fsqrt_rn(v_lo) * (1.0f - 4.0f * kU);
            bool bail = !(v_hi < 3.0e38f);

            // ---- exact median (qsort.go:68-82) ----
            const int kk = a + (cnt >> 1);
            const float upper = pick_rank<LPP, NS, TOPW, BOTW>(v, kk, role, MIDR);
            const float lower = pick_rank<LPP, NS, TOPW, BOTW>(v, kk - 1, role, MIDR);
            const float median = (cnt & 1) ? upper : 0.5f * (lower + upper);

            if constexpr (WINSOR) {
                // ---- winsorized stddev (stack.go:646-672) as an interval, see
                // WinsorInterval in fast_common.hpp.  Every lane clamps its own NS
                // ranks; the four partial sums meet in quad_sum. ----
                const float xmin = ZONAL ? pick_rank<LPP, NS, 1, ZL>(v, a, role, -1)
                                         : pick_rank<LPP, NS, NS, NS>(v, a, role, 0);
                const float xmax = ZONAL ? pick_rank<LPP, NS, ZHS, 1>(v, b - 1, role, LAST)
                                         : pick_rank<LPP, NS, NS, NS>(v, b - 1, role, 0);
                WinsorInterval wi;
                wi.start(s_min, s_max);
                const float inv_cnt = 1.0f / fcnt;
                bool inner = active && !bail;
                // ranks below a / from b on are excluded: only lane 0 / the last lane see them
                const int a_loc = role == 0 ? a : (role > LAST ? NS : 0);
                const int b_loc = role == LAST ? b - LAST * NS : (role > LAST ? 0 : NS);
                while (__any(inner)) {
                    if (ZONAL) { if (lane == 0) NL_STAT(0, 1); if (inner && role == 0) NL_STAT(1, 1); }
                    // (re-materialised per round: otherwise one lane mask per position is kept in SGPRs)
                    const int al = opaque(a_loc), bl = opaque(b_loc);
                    wi.next_clamp(median, xmin, xmax);
                    auto clamped_variance = [&](const float Lt, const float Ht, float &wvar, float &werr, float &wmean_c,
                                                float &wrms) NL_INL {
                        float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                        if constexpr (ZONAL) {
                            static_range<0, ZL>([&](auto K) NL_INL {
                                constexpr int k = decltype(K)::value;
                                const float e = (k >= al) ? __builtin_amdgcn_fmed3f(v[k], Lt, Ht) - cz : 0.0f;
                                d0 += e; q0 = __builtin_fmaf(e, e, q0);
                            });
                            static_chunks<0, (NS - ZHS - ZL) / 4, 4>([&](auto K) NL_INL {
                                constexpr int k = ZL + 4 * decltype(K)::value;
                                const float e0 = __builtin_amdgcn_fmed3f(v[k], Lt, Ht) - cz;
                                const float e1 = __builtin_amdgcn_fmed3f(v[k + 1], Lt, Ht) - cz;
                                const float e2 = __builtin_amdgcn_fmed3f(v[k + 2], Lt, Ht) - cz;
                                const float e3 = __builtin_amdgcn_fmed3f(v[k + 3], Lt, Ht) - cz;
                                d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                                q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                                q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
                            });
                            static_range<NS - ZHS, NS>([&](auto K) NL_INL {
                                constexpr int k = decltype(K)::value;
                                const float e = (k < bl) ? __builtin_amdgcn_fmed3f(v[k], Lt, Ht) - cz : 0.0f;
                                d1 += e; q1 = __builtin_fmaf(e, e, q1);
                            });
                        } else {
                            const int a4 = opaque(a - role * NS);
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
                        const float wd = quad_sum<LPP>((d0 + d1) + (d2 + d3)) * inv_cnt;
                        const float wa = quad_sum<LPP>((q0 + q1) + (q2 + q3)) * inv_cnt;
                        const float wb = wd * wd;
                        wvar = fmaxf(wa - wb, 0.0f);
                        werr = ((float)(NS / 2 + 48)) * kU * (wa + wb);
                        wmean_c = wd;
                        wrms = wa;
                    };
                    float var_t, err_t, wd_t, wa_t;
                    clamped_variance(wi.Lp, wi.Hm, var_t, err_t, wd_t, wa_t);
                    // loosest clamp (Lm, Hp): first-order bound from the number of clamped samples instead
                    // of a second evaluation (see stack_fast.hip); every lane is fully sorted, so testing
                    // every 8th rank bounds the counts to +7
                    int t_lo = 0, t_hi = 0;
                    if constexpr (ZONAL) {
                        static_range<0, NS / 8>([&](auto J) NL_INL {
                            constexpr int k = 8 * decltype(J)::value;
                            const bool lo_in = (k + 7 >= ZL) || (k + 7 >= al);
                            const bool hi_in = (k < NS - ZHS) || (k < bl);
                            t_lo += (lo_in && v[k + 7] < wi.Lp) ? 1 : 0;
                            t_hi += (hi_in && v[k] > wi.Hm) ? 1 : 0;
                        });
                        if (role > LAST) { t_lo = 0; t_hi = 0; }        // (WIDE) lanes of padding
                    } else {
                        const int a5 = opaque(a - role * NS);
                        static_range<0, NS / 8>([&](auto J) NL_INL {
                            constexpr int k = 8 * decltype(J)::value;
                            const bool lo_in = (unsigned)(k + 7 - a5) < (unsigned)cnt;
                            const bool hi_in = (unsigned)(k - a5) < (unsigned)cnt;
                            t_lo += (lo_in && v[k + 7] < wi.Lp) ? 1 : 0;
                            t_hi += (hi_in && v[k] > wi.Hm) ? 1 : 0;
                        });
                    }
                    t_lo = quad_sum<LPP>(t_lo);
                    t_hi = quad_sum<LPP>(t_hi);
                    float var_l, err_l;
                    {
                        const float n_lo = (float)min(8 * t_lo + 7, cnt), n_hi = (float)min(8 * t_hi + 7, cnt);
                        const float dL = (wi.Lp - wi.Lm) * (1.0f + 2.0f * kU), dH = (wi.Hp - wi.Hm) * (1.0f + 2.0f * kU);
                        const float ybar = cz + wd_t;
                        const float slop = 4.0e-6f * __builtin_amdgcn_sqrtf(wa_t) + 4.0f * kU * fabsf(ybar) + 1.0e-30f;
                        const float gL = fmaxf(ybar - wi.Lp, 0.0f) + slop, gH = fmaxf(wi.Hm - ybar, 0.0f) + slop;
                        const float corr = (n_lo * (dL * (2.0f * gL + dL)) + n_hi * (dH * (2.0f * gH + dH))) * inv_cnt;
                        var_l = var_t + ((corr == corr) ? corr * 1.001f : 0.0f);
                        err_l = err_t;
                    }
                    wi.finish_round(var_t, err_t, var_l, err_l, eps_r, e_m, true, inner, bail);
                }
                s_min = wi.hull_lo;
                s_max = wi.hull_hi;
            }

            // ---- the reference's bound expressions (stack.go:408-409) at both ends ----
            const float tl0 = __fmul_rn(p.sig_lo, s_min), tl1 = __fmul_rn(p.sig_lo, s_max);
            const float th0 = __fmul_rn(p.sig_hi, s_min), th1 = __fmul_rn(p.sig_hi, s_max);
            const float la = __fsub_rn(median, tl0), lb = __fsub_rn(median, tl1);
            const float ha = __fadd_rn(median, th0), hb = __fadd_rn(median, th1);
            const float lo_min = fminf(la, lb), lo_max = fmaxf(la, lb);
            const float hi_min = fminf(ha, hb), hi_max = fmaxf(ha, hb);

            // ---- certain (c1,d1) and possible (c2,d2) clips: sorted => prefix / suffix ----
            int c1 = 0, c2 = 0, d1 = 0, d2 = 0;
            if constexpr (ZONAL) {
                static_range<0, ZL>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    c1 += (v[k] < lo_min) ? 1 : 0;
                    c2 += (v[k] < lo_max) ? 1 : 0;
                });
                static_range<NS - ZHS, NS>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    d1 += (v[k] > hi_max) ? 1 : 0;
                    d2 += (v[k] > hi_min) ? 1 : 0;
                });
                c1 = quad_sum<LPP>(role == 0 ? c1 : 0); c2 = quad_sum<LPP>(role == 0 ? c2 : 0);
                d1 = quad_sum<LPP>(role == LAST ? d1 : 0); d2 = quad_sum<LPP>(role == LAST ? d2 : 0);
                c1 = max(c1 - a, 0); c2 = max(c2 - a, 0);
                d1 = max(d1 - (NTE - b), 0); d2 = max(d2 - (NTE - b), 0);
                if (active && ((a + c2 >= ZL) || (b - d2 <= ZH))) {
                    to_generic = true;
                    active = false;
                }
            } else {
                static_chunks<0, NS, 8>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const float x = v[k];
                    c1 += (x < lo_min) ? 1 : 0;
                    c2 += (x < lo_max) ? 1 : 0;
                    d1 += (x > hi_max) ? 1 : 0;
                    d2 += (x > hi_min) ? 1 : 0;
                });
                c1 = quad_sum<LPP>(c1); c2 = quad_sum<LPP>(c2);
                d1 = quad_sum<LPP>(d1); d2 = quad_sum<LPP>(d2);
                c1 = min(max(c1 - a, 0), cnt); c2 = min(max(c2 - a, 0), cnt);
                d1 = min(max(d1 - (NT - b), 0), cnt); d2 = min(max(d2 - (NT - b), 0), cnt);
            }
            if (active) {
                bail |= (c1 != c2) || (d1 != d2) || (lo_max > hi_min && (c1 + d1) > 0);
                if (bail) {
                    to_exact = true;
                    active = false;
                } else {
                    c_lo += c1;
                    c_hi += d1;
                    a += c1;
                    b -= d1;
                    amax = fminf(amax, fmaxf(fabsf(lo_min), fabsf(hi_max)));
                    if ((c1 + d1) == 0 || (b - a) <= 1) {     // stack.go:427-430
                        res = m;
                        active = false;
                    }
                }
            }
        }

        // one lane per pixel reports
        const bool rep = on && role == 0;
        if (rep && !to_generic && !to_exact) {
            p.out[pix] = res;
            c_lo_total += c_lo;
            c_hi_total += c_hi;
        }
        if constexpr (ZONAL) {
            const unsigned long long gm = __ballot(rep && to_generic);
            if (gm) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(q.gen_count, (unsigned)__popcll(gm));
                base = __shfl(base, 0, 64);
                const unsigned slot = base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull));
                if (rep && to_generic && slot < q.gen_capacity) q.gen_list[slot] = (unsigned)pix;
            }
        }
        const unsigned long long em = __ballot(rep && to_exact);
        if (em) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
            base = __shfl(base, 0, 64);
            const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
            if (rep && to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
        }
    }

    __shared__ int s_lo[4], s_hi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_lo_total += __shfl_xor(c_lo_total, o, 64);
        c_hi_total += __shfl_xor(c_hi_total, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = c_lo_total; s_hi[threadIdx.x >> 6] = c_hi_total; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t_lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
        const int t_hi = s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3];
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}
#endif  // NL_EXPERIMENTS

// StackMedian (stack.go:274-303) for 129..512 frames: the merged column gives the median
// exactly (order independent); both middle ranks are looked up over whole lanes, so any
// number of missing samples is fine and nothing is handed over.
template <int LPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void stack_median_ml_kernel(StackArgs p)
{
    constexpr int NS = kMlNS;
    const int role = threadIdx.x % LPP;
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x / LPP) + threadIdx.x / LPP;
    const bool on = item < p.npix;
    int N = p.n_frames;
    asm volatile("" : "+s"(N));
    float v[NS];
    const int n = ml_gather_sorted<LPP, NS, false, 16, 32, NS, NS / 2, NS, LPP == 2>(p.frames, p.stride, N, on, item, role, v);      // (nt loads at two lanes per pixel: fast_ml_common.hpp)
    const int kk = n >> 1;
    const float upper = pick_rank<LPP, NS, NS, NS>(v, kk, role, 0);
    const float lower = pick_rank<LPP, NS, NS, NS>(v, kk > 0 ? kk - 1 : 0, role, 0);
    float res = (n & 1) ? upper : 0.5f * (lower + upper);          // qsort.go:73-81
    if (n == 0) res = p.ref_loc;
    if (on && role == 0) NL_STORE_RESULT(&p.out[item], res);
}

hipError_t launch_stack_median_ml(const StackArgs &args, hipStream_t stream, const char **name)
{
    if (args.n_frames <= 2 * kMlNS) {
        *name = "stack_median_ml_kernel<2>";
        const unsigned blocks = (unsigned)((args.npix + 127) / 128);
        hipLaunchKernelGGL(stack_median_ml_kernel<2>, dim3(blocks), dim3(256), 0, stream, args);
    } else {
        *name = "stack_median_ml_kernel<4>";
        const unsigned blocks = (unsigned)((args.npix + 63) / 64);
        hipLaunchKernelGGL(stack_median_ml_kernel<4>, dim3(blocks), dim3(256), 0, stream, args);
    }
    return hipGetLastError();
}

// StackMADSigma (stack.go:536-605) for 129..512 frames, as stack_mad_fast_kernel: merged
// column -> median; column := |x - median| -> bitonic merge -> MAD; second read of the
// pixel's frames (every lane its own) for the clip counts and the mean of the survivors.
template <int LPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void stack_mad_ml_kernel(StackArgs p, FastArgs q)
{
    constexpr int NS = kMlNS;
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x % LPP;
    const int64_t pix = (int64_t)blockIdx.x * (blockDim.x / LPP) + threadIdx.x / LPP;
    const bool on = pix < p.npix;
    int N = p.n_frames;
    asm volatile("" : "+s"(N));
    float v[NS];
    const int n = ml_gather_sorted<LPP, NS, false>(p.frames, p.stride, N, on, pix, role, v);
    const int kk = n >> 1;
    const float upper = pick_rank<LPP, NS, NS, NS>(v, kk, role, 0);
    const float lower = pick_rank<LPP, NS, NS, NS>(v, kk > 0 ? kk - 1 : 0, role, 0);
    const float median = (n & 1) ? upper : 0.5f * (lower + upper);
    const bool degenerate = n > 0 && !(__builtin_fabsf(median) < __builtin_inff());
    const float msafe = degenerate ? 0.0f : median;
    static_chunks<0, NS, 16>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        v[k] = __builtin_fabsf(v[k] - msafe);
    });
    // The deviations of the sorted column (lane r: ranks [128 r, 128 r + 128)) fall to the median and rise
    // again, +Inf pads on top: ONE bitonic sequence over the lanes of the pixel.  The half-cleaner cascade
    // of a bitonic merge sorts it -- lane distance 2 and 1 (element i meets the partner's element i), then
    // the in-lane cascade -- without the 2 184-operation sort of every lane and the run merges.
    if constexpr (LPP == 4) cross_stage<NS, kSwap2, false>(v, role < 2);
    cross_stage<NS, kSwap1, false>(v, (role & 1) == 0);
    run_network<FusedBitonic<NS, 0>, NS>(v);
    const float dupper = pick_rank<LPP, NS, NS, NS>(v, kk, role, 0);
    const float dlower = pick_rank<LPP, NS, NS, NS>(v, kk > 0 ? kk - 1 : 0, role, 0);
    const float mad = (n & 1) ? dupper : 0.5f * (dlower + dupper);
    const float sd = mad * 1.4826f;
    const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
    const float lo = median - t_lo, hi = median + t_hi;

    // second read: lane role r takes frames r, r+LPP, ... again (clipped descriptors as in the gather)
    int N2 = p.n_frames;
    asm volatile("" : "+s"(N2));
    int frame_bytes = (int)(p.stride * (int64_t)sizeof(float));
    asm volatile("" : "+s"(frame_bytes));
    const int voff = (int)((unsigned)(on ? pix : 0) * 4u) + role * frame_bytes;
    static_chunks<0, NS, 4>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        const int avail = min(max(N2 - k * LPP, 0), LPP);
        const char *gb = reinterpret_cast<const char *>(p.frames) + (int64_t)(k * LPP) * frame_bytes;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gb), 0, avail * frame_bytes, 0x00020000);
        v[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
    });
    float f_lo = 0.0f, f_hi = 0.0f, f_kept = 0.0f, sum = 0.0f;
    static_chunks<0, NS, 4>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        const float x = v[k];
        const bool present = (k * LPP + role < N2) && (x == x);
        const bool below = present && x < lo;
        const bool above = present && !below && x > hi;
        const bool keep = present && !below && !above;
        f_lo += below ? 1.0f : 0.0f;
        f_hi += above ? 1.0f : 0.0f;
        f_kept += keep ? 1.0f : 0.0f;
        sum += keep ? x : 0.0f;
        asm volatile("" : "+v"(f_lo), "+v"(f_hi), "+v"(f_kept), "+v"(sum));
    });
    int c_lo = (int)quad_sum<LPP>(f_lo), c_hi = (int)quad_sum<LPP>(f_hi);
    float res = quad_sum<LPP>(sum) / quad_sum<LPP>(f_kept);
    if (n == 0) res = p.ref_loc;
    const bool rep = on && role == 0;
    const bool to_exact = rep && degenerate;
    if (rep && !to_exact) NL_STORE_RESULT(&p.out[pix], res);
    if (!rep || to_exact || n == 0) { c_lo = 0; c_hi = 0; }
    const unsigned long long em = __ballot(to_exact);
    if (em) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
        if (to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
    }
    __shared__ int s_lo[4], s_hi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_lo += __shfl_xor(c_lo, o, 64);
        c_hi += __shfl_xor(c_hi, o, 64);
    }
    if (lane == 0) { s_lo[threadIdx.x >> 6] = c_lo; s_hi[threadIdx.x >> 6] = c_hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t_l = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
        const int t_h = s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3];
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (t_l) atomicAdd(slot + 0, (unsigned long long)t_l);
        if (t_h) atomicAdd(slot + 1, (unsigned long long)t_h);
    }
}

hipError_t launch_stack_mad_ml(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name)
{
    if (args.n_frames <= 2 * kMlNS) {
        *name = "stack_mad_ml_kernel<2>";
        hipLaunchKernelGGL(stack_mad_ml_kernel<2>, dim3((unsigned)((args.npix + 127) / 128)), dim3(256), 0, stream, args,
                           fargs);
    } else {
        *name = "stack_mad_ml_kernel<4>";
        hipLaunchKernelGGL(stack_mad_ml_kernel<4>, dim3((unsigned)((args.npix + 63) / 64)), dim3(256), 0, stream, args,
                           fargs);
    }
    return hipGetLastError();
}

int fast_ml_supported(int mode, bool weighted, int n_frames, int64_t npix)
{
    // 4 frames of the tile must be addressable with a 31-bit buffer offset
    if (n_frames <= 128 || n_frames > 512 || npix >= ((int64_t)1 << 27)) return 0;
    if (mode == NL_ST_MEDIAN) return 1;
    if (mode == NL_ST_MAD_SIGMA) return weighted ? 0 : 1;
    return ((mode == NL_ST_SIGMA || mode == NL_ST_WINSOR_SIGMA) && !weighted) ? 1 : 0;
}

// the nested launchers end in hipGetLastError(), which clears the pending error: keep the first one
static inline void keep_first(hipError_t &acc, hipError_t e) { if (acc == hipSuccess) acc = e; }

// Dominant kernel = the LDS-column kernel of the frame-count class (stack_fast_mlz*.hip), generic pass = whole columns in
// LDS (stack_fast_mlg.hip).  The experiments build (make EXPERIMENTS=1) can put the round-1 register-zone kernel back
// in either place: NL_MLZ=0 / NL_MLG=0.
template <int LPP, bool WINSOR, bool WIDE>
static hipError_t launch_ml(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                            hipEvent_t dominant_done, AfterDominant after, void *user, const char **mlz_name, hipStream_t tail)
{
    hipError_t err = hipSuccess;
    FastArgs f = fargs;
    f.in_list = nullptr;
    f.in_count = nullptr;
    f.in_capacity = 0;
#ifdef NL_EXPERIMENTS
    const unsigned per_wg = 256 / LPP;
    const unsigned tile_blocks = (unsigned)((args.npix + per_wg - 1) / per_wg);
    if (WIDE || !mlz_name) {
        hipLaunchKernelGGL((stack_sigma_ml_kernel<LPP, true, WINSOR, WIDE>), dim3(tile_blocks), dim3(256), 0, stream,
                           args, f);
        keep_first(err, hipGetLastError());
    } else
#endif
    {
        // every position in use: the clipping rounds run on LDS columns (stack_fast_mlz.hip)
        keep_first(err, launch_stack_sigma_mlz(args, f, stream, mlz_name, WINSOR));
    }
    if (dominant_done) keep_first(err, hipEventRecord(dominant_done, stream));
    if (after) after(user);
    if (tail) stream = tail;          // chunked passes: the generic pass on a stream of its own (the callback ordered it)
    f.in_list = fargs.gen_list;
    f.in_count = fargs.gen_count;
    f.in_capacity = fargs.gen_capacity;
#ifdef NL_EXPERIMENTS
    static const bool mlg_on = [] { const char *e = getenv("NL_MLG"); return !(e && e[0] == '0'); }();
    if (!mlg_on) {
        const unsigned gblocks = tile_blocks < kGenericGrid ? tile_blocks : kGenericGrid;
        hipLaunchKernelGGL((stack_sigma_ml_kernel<LPP, false, WINSOR, false>), dim3(gblocks), dim3(256), 0, stream,
                           args, f);
        keep_first(err, hipGetLastError());
        return err;
    }
#endif
    // generic pass over the hand-over list: whole columns in LDS (stack_fast_mlg.hip)
    keep_first(err, launch_stack_sigma_mlg(args, f, generic_grid(fargs.gen_hint, 64 / LPP, 4 * kGenericGrid), stream, WINSOR));
    return err;
}

template <int LPP>
static hipError_t launch_ml_variant(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name,
                              hipEvent_t dominant_done, bool winsor, AfterDominant after, void *user, hipStream_t tail)
{
    const int n = args.n_frames;
#ifdef NL_EXPERIMENTS
    // tight zones when (almost) every position is used, otherwise the wide variant while the
    // last lane with samples holds more than 8 of them; in the few remaining cases the tight
    // variant hands every pixel to the generic pass
    const int nt = LPP * kMlNS;
    const bool wide = n < nt - 8 && n > ((n - 1) / kMlNS) * kMlNS + 8;
    static const std::string names[4] = {
        "stack_sigma_ml_kernel<" + std::to_string(LPP) + ", true, false, false>",
        "stack_sigma_ml_kernel<" + std::to_string(LPP) + ", true, false, true>",
        "stack_sigma_ml_kernel<" + std::to_string(LPP) + ", true, true, false>",
        "stack_sigma_ml_kernel<" + std::to_string(LPP) + ", true, true, true>"};
    *name = names[(winsor ? 2 : 0) + (wide ? 1 : 0)].c_str();
    static const bool mlz_on = [] { const char *e = getenv("NL_MLZ"); return !(e && e[0] == '0'); }();
    if (!(mlz_on && fast_mlz_supported(winsor ? NL_ST_WINSOR_SIGMA : NL_ST_SIGMA, false, n))) {
        if (winsor) {
            if (wide) return launch_ml<LPP, true, true>(args, fargs, stream, dominant_done, after, user, nullptr, tail);
            return launch_ml<LPP, true, false>(args, fargs, stream, dominant_done, after, user, nullptr, tail);
        }
        if (wide) return launch_ml<LPP, false, true>(args, fargs, stream, dominant_done, after, user, nullptr, tail);
        return launch_ml<LPP, false, false>(args, fargs, stream, dominant_done, after, user, nullptr, tail);
    }
#else
    (void)n;
#endif
    // every frame count 129..512: LDS-column kernel of its class (stack_fast_mlz.hip), which sets *name
    if (winsor) return launch_ml<LPP, true, false>(args, fargs, stream, dominant_done, after, user, name, tail);
    return launch_ml<LPP, false, false>(args, fargs, stream, dominant_done, after, user, name, tail);
}

// kernel names as rocprofv3 prints them (template arguments: LPP, ZONAL, WINSOR, WIDE)
hipError_t launch_stack_sigma_ml(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                 const char **name, hipEvent_t dominant_done, bool winsor,
                                 AfterDominant after, void *user, hipStream_t tail)
{
    hipError_t err = args.n_frames <= 2 * kMlNS
        ? launch_ml_variant<2>(args, fargs, stream, name, dominant_done, winsor, after, user, tail)
        : launch_ml_variant<4>(args, fargs, stream, name, dominant_done, winsor, after, user, tail);
    keep_first(err, hipGetLastError());
    return err;
}

}  // namespace nl
