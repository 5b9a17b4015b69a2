// stack_fast_mlg.hip -- the GENERIC pass of the multi-lane sigma / winsorized sigma kernels
// (129..512 frames; 65..128 frames for winsorized clipping): pixels a zonal kernel (stack_fast.hip,
// stack_fast_ml.hip, stack_fast_mlz.hip) handed over
// because they miss too many samples (the NaN borders of aligned frames) or clip / clamp more
// samples than its zones hold.
//
// As stack_fast_mlz.hip, but with the WHOLE merged column of a pixel in LDS ([rank][pixel],
// conflict-free) next to prefix sums of (x-c) and (x-c)^2 at every 4th rank: the alive window
// [a, b), the clamp positions and the median are plain indices into the column, any number of
// missing, clipped or clamped samples, and a round costs a few dozen LDS reads whatever the frame
// count.  (The register version of this pass masks all 128 positions of every lane in every
// round and runs one wave per SIMD: 1.9 ms for the 34 k border pixels of the C3 tile, half of
// that pass.)  One wave per workgroup, 48 KiB of LDS: three waves per CU -- this kernel only ever
// sees hand-over lists.
//
// The prefix sums run upwards from rank 0, so a sum over [i, j) is a difference of two entries
// that both contain every sample below i -- dead low outliers included.  The rounding bound
// therefore uses the magnitude of the prefix itself (`mag` below) instead of the alive samples'
// moments: a pixel with a deep cold outlier gets a wider interval, never a wrong decision.
// Exactness otherwise as in stack_fast.hip / DESIGN.md section 5.
#include "fast_ml_common.hpp"

namespace nl {

#ifdef NL_ROUND_STATS
__device__ unsigned long long nl_dbg_rounds_mlg[8];          // as nl_dbg_rounds in stack_fast.hip; [5] walk steps
extern "C" int nl_debug_round_stats_mlg(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nl_dbg_rounds_mlg), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nl_dbg_rounds_mlg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define NL_STAT(i, x) atomicAdd(&nl_dbg_rounds_mlg[i], (unsigned long long)(x))
#else
#define NL_STAT(i, x) ((void)0)
#endif

namespace {

template <int LPP>
struct MlgLayout {
    static constexpr int NS = kMlNS, NT = NS * LPP;
    static constexpr int PW = 64 / LPP;                         // pixels per wave = LDS row length
    static constexpr int G = NT / 4;                            // prefix entries 0 .. G
    static constexpr int X = 0, P1 = NT, P2 = P1 + G + 1, ROWS = P2 + G + 1;
    // roundings a term can see: chain inside a lane, lane offset, table difference, partial groups, assembly
    static constexpr int ROUNDINGS = NS + 20;
};

__device__ __forceinline__ void lds_settle_g() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// first rank j' in [j, lim) whose sample is not below thr (sorted column); COARSE: strides of 8 first
template <int PW, bool COARSE>
__device__ __forceinline__ int walk_up(const float *x, int j, int lim, float thr, bool on, int top)
{
    if (COARSE) {
        bool more = on;
        while (__any(more)) {
            int t = 0;
            static_range<0, 8>([&](auto S) NL_INL {
                const int idx = j + 8 * decltype(S)::value + 7;
                const float val = x[min(idx, top) * PW];
                t += (idx < lim && val < thr) ? 1 : 0;
            });
            if (more) { j += 8 * t; more = t == 8; }
        }
    }
    bool more = on;
    while (__any(more)) {
        int t = 0;
        static_range<0, 8>([&](auto I) NL_INL {
            const int idx = j + decltype(I)::value;
            const float val = x[min(idx, top) * PW];
            t += (idx < lim && val < thr) ? 1 : 0;
        });
        if (more) { j += t; more = t == 8; }
    }
    return j;
}

// smallest j' in [lim, j] such that every rank in [j', j) is above thr
template <int PW, bool COARSE>
__device__ __forceinline__ int walk_down(const float *x, int j, int lim, float thr, bool on)
{
    if (COARSE) {
        bool more = on;
        while (__any(more)) {
            int t = 0;
            static_range<0, 8>([&](auto S) NL_INL {
                const int idx = j - 8 * decltype(S)::value - 8;
                const float val = x[max(idx, 0) * PW];
                t += (idx >= lim && val > thr) ? 1 : 0;
            });
            if (more) { j -= 8 * t; more = t == 8; }
        }
    }
    bool more = on;
    while (__any(more)) {
        int t = 0;
        static_range<0, 8>([&](auto I) NL_INL {
            const int idx = j - 1 - decltype(I)::value;
            const float val = x[max(idx, 0) * PW];
            t += (idx >= lim && val > thr) ? 1 : 0;
        });
        if (more) { j -= t; more = t == 8; }
    }
    return j;
}

// sum of (x-c) and (x-c)^2 over the ranks [i, j) of the column, 0 <= i <= j: prefix tables plus the
// partial groups at both ends; pmag = prefix of the squares up to j (the magnitude the roundings scale with)
template <class LY>
__device__ __forceinline__ void range_moments(const float *col, float c, int i, int j, int top, float &d, float &q,
                                              float &pmag)
{
    constexpr int PW = LY::PW;
    const int gi = (i + 3) >> 2, gj = j >> 2;
    const bool tabled = gi <= gj;
    const float p1i = col[(LY::P1 + gi) * PW], p1j = col[(LY::P1 + gj) * PW];
    const float p2i = col[(LY::P2 + gi) * PW], p2j = col[(LY::P2 + gj) * PW];
    // partial groups: ranks [i, min(4 gi, j)) and, when the tables are used, [4 gj, j)
    const int lo_end = tabled ? 4 * gi : j;
    const int hi_beg = tabled ? 4 * gj : j;
    float dp = 0.0f, qp = 0.0f;
    static_range<0, 3>([&](auto U) NL_INL {
        constexpr int u = decltype(U)::value;
        const int kl = i + u, kh = j - 1 - u;
        const float xl = col[(LY::X + min(kl, top)) * PW], xh = col[(LY::X + max(kh, 0)) * PW];
        const float e = (kl < lo_end) ? xl - c : 0.0f;
        const float f = (kh >= hi_beg) ? xh - c : 0.0f;
        dp += e; qp = __builtin_fmaf(e, e, qp);
        dp += f; qp = __builtin_fmaf(f, f, qp);
    });
    d = dp + (tabled ? p1j - p1i : 0.0f);
    q = qp + (tabled ? p2j - p2i : 0.0f);
    pmag = (tabled ? p2j : 0.0f) + qp;
}

}  // namespace

// The kernel's body, workgroup `block` of `nblocks` (stack_sigma_mlg_kernel below: the whole grid; stack_tail_fused.hip: the
// upper workgroups of a grid whose lower part replays the dominant kernel's exact list).
template <int LPP, bool WINSOR>
__device__ __forceinline__ void mlg_body(const StackArgs &p, const FastArgs &q, const unsigned block, const unsigned nblocks)
{
    using LY = MlgLayout<LPP>;
    constexpr int NS = LY::NS, NT = LY::NT, PW = LY::PW;
    __shared__ float lds[LY::ROWS * PW];
    if (q.in_list) { fused_collect_slots(p, block); snapshot_fb_list(q); }

    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x % LPP;
    float *col = lds + threadIdx.x / LPP;                  // element r of this pixel: col[r * PW]
    constexpr int top = NT - 1;
    constexpr float kErrF = (float)(2 * LY::ROUNDINGS + 8);

    int c_lo_total = 0, c_hi_total = 0;
    const int64_t limit = q.in_list ? (int64_t)min(*q.in_count, q.in_capacity) : p.npix;
    const int64_t sweep = (int64_t)nblocks * PW;

    for (int64_t wg_item = (int64_t)block * PW; wg_item < limit; wg_item += sweep) {
        int N = p.n_frames;
        asm volatile("" : "+s"(N));
        const int64_t item = wg_item + threadIdx.x / LPP;
        const bool on = item < limit;
        int64_t pix = item;
        if (q.in_list) pix = on ? (int64_t)q.in_list[item] : 0;

#ifdef NL_ROUND_STATS
        const unsigned long long t0 = __builtin_readcyclecounter();
#endif
        float v[NS];
        int n;
        if constexpr (LPP == 1) n = gather_sorted<NS, NS>(p.frames, p.stride, N, (unsigned)(on ? pix : 0) * 4u, v);
        else                    n = ml_gather_sorted<LPP, NS, false>(p.frames, p.stride, N, on, pix, role, v);

#ifdef NL_ROUND_STATS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
#endif
        // ---- the whole column to LDS: lane r holds ranks [r NS, r NS + NS) ----
        {
            float *mine = col + (LY::X + role * NS) * PW;
            static_range<0, NS>([&](auto K) NL_INL { mine[decltype(K)::value * PW] = v[decltype(K)::value]; });
        }
        lds_settle_g();
        bool active = on && n > 0;
        bool to_exact = false;
        const float c = col[(LY::X + min(max(n >> 1, 0), top)) * PW];       // shift: the first median

        // ---- prefix sums of (x-c), (x-c)^2 at every 4th rank: lane totals, then one chain per lane ----
        {
            float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            static_chunks<0, NS / 4, 4>([&](auto K) NL_INL {
                constexpr int k = 4 * decltype(K)::value;
                const float e0 = v[k] - c, e1 = v[k + 1] - c, e2 = v[k + 2] - c, e3 = v[k + 3] - c;
                d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
            });
            const float t1 = (d0 + d1) + (d2 + d3), t2 = (q0 + q1) + (q2 + q3);
            // totals of the lanes below this one (missing samples = +Inf only reach entries above n, never read)
            float o1 = 0.0f, o2 = 0.0f;
            if constexpr (LPP > 1) {
                const float a1 = dpp_f<kSwap1>(t1), a2 = dpp_f<kSwap1>(t2);       // partner lane ^ 1
                if constexpr (LPP == 2) {
                    o1 = (role & 1) ? a1 : 0.0f;
                    o2 = (role & 1) ? a2 : 0.0f;
                } else {
                    const float pair1 = t1 + a1, pair2 = t2 + a2;                 // lanes {0,1} or {2,3}
                    const float b1 = dpp_f<kSwap2>(pair1), b2 = dpp_f<kSwap2>(pair2);
                    o1 = ((role & 1) ? a1 : 0.0f) + ((role & 2) ? b1 : 0.0f);
                    o2 = ((role & 1) ? a2 : 0.0f) + ((role & 2) ? b2 : 0.0f);
                }
            }
            float *m1 = col + (LY::P1 + role * (NS / 4)) * PW, *m2 = col + (LY::P2 + role * (NS / 4)) * PW;
            float s1 = o1, s2 = o2;
            static_range<0, NS / 4>([&](auto G) NL_INL {
                constexpr int g = decltype(G)::value;
                m1[g * PW] = s1;                           // prefix below rank role NS + 4g
                m2[g * PW] = s2;
                static_range<0, 4>([&](auto U) NL_INL {
                    const float e = v[4 * g + decltype(U)::value] - c;
                    s1 += e;
                    s2 = __builtin_fmaf(e, e, s2);
                });
            });
            if (role == LPP - 1) {                         // entry G (the whole column); after the column is dead
                m1[(NS / 4) * PW] = s1;
                m2[(NS / 4) * PW] = s2;
            }
        }
        lds_settle_g();

#ifdef NL_ROUND_STATS
        const unsigned long long t2 = __builtin_readcyclecounter();
#endif
        float res = p.ref_loc;
        int c_lo = 0, c_hi = 0;
        int a = 0, b = n;                                  // survivors = sorted ranks [a, b)
        const float *x = col + LY::X * PW;

        if (lane == 0) NL_STAT(4, 1);
        while (__any(active)) {
            if (lane == 0) NL_STAT(2, 1);
            if (active && role == 0) NL_STAT(3, 1);
            const int cnt = b - a;
            const float fcnt = (float)cnt;
            const float inv_cnt = 1.0f / fcnt;
            const int kk = min(max(a + (cnt >> 1), 0), top);       // (a single survivor: rank a itself)
            const float upper = x[kk * PW], lower = x[max(kk - 1, 0) * PW];
            const float median = (cnt & 1) ? upper : 0.5f * (lower + upper);       // qsort.go:68-82
            const float xmin = x[min(max(a, 0), top) * PW], xmax = x[min(max(b - 1, 0), top) * PW];

            float dsum, qsum, pmag;
            range_moments<LY>(col, c, a, b, top, dsum, qsum, pmag);
            const float delta = dsum * inv_cnt;            // mean~ - c
            const float m = c + delta;
            const float aa = qsum * inv_cnt;               // E[(x-c)^2]~
            const float bb = delta * delta;
            const float var = fmaxf(aa - bb, 0.0f);
            // roundings scale with the prefix up to b, spread over cnt survivors (see the file header)
            const float mag = pmag * inv_cnt * ((float)b * inv_cnt) + bb;

            // ---- bracket the reference's stddev ----
            const float amax = fmaxf(fabsf(xmin), fabsf(xmax));
            const float err_o = kErrF * kU * mag;
            const float eps_r = 1.02f * (fcnt + 8.0f) * kU;
            const float e_m = 1.02f * (fcnt + 2.0f) * kU * amax;
            const float v_up = var + err_o;
            const float v_dn = fmaxf(var - err_o, 0.0f);
            const float v_hi = v_up + v_up * eps_r + e_m * e_m;
            const float v_lo = fmaxf(v_dn - v_dn * eps_r, 0.0f);
            float s_max = __fsqrt_rn(v_hi) * (1.0f + 4.0f * kU);
            float s_min = __fsqrt_rn(v_lo) * (1.0f - 4.0f * kU);
            bool bail = !(v_hi < 3.0e38f);

            if constexpr (WINSOR) {
                // ---- winsorized stddev (stack.go:646-672) as an interval, WinsorInterval in fast_common.hpp ----
                // ranks [a, jl) sit on the low clamp, [jh, b) on the high clamp; both only tighten inside one loop
                WinsorInterval wi;
                wi.start(s_min, s_max, q.gen_round_cap > 0 ? q.gen_round_cap : 100);      // (see FastArgs::gen_round_cap)
                bool inner = active && !bail;
                int jl = a, jh = b;
                bool first = true;
                while (__any(inner)) {
                    if (lane == 0) NL_STAT(0, 1);
                    if (inner && role == 0) NL_STAT(1, 1);
                    wi.next_clamp(median, xmin, xmax);
                    if (first) {
                        jl = walk_up<PW, true>(x, jl, b, wi.Lp, inner, top);
                        jh = walk_down<PW, true>(x, jh, jl, wi.Hm, inner);
                        first = false;
                    } else {
                        jl = walk_up<PW, false>(x, jl, jh, wi.Lp, inner, top);
                        jh = walk_down<PW, false>(x, jh, jl, wi.Hm, inner);
                    }
                    float du, qu, pm;
                    range_moments<LY>(col, c, jl, jh, top, du, qu, pm);
                    const float n_lo = (float)(jl - a), n_hi = (float)(b - jh);
                    const float eL = wi.Lp - c, eH = wi.Hm - c;               // max(x, Lp) - c of a clamped sample
                    const float dcl = n_lo * eL + n_hi * eH;
                    const float qcl = n_lo * (eL * eL) + n_hi * (eH * eH);
                    const float wd = (du + dcl) * inv_cnt;
                    const float wa = (qu + qcl) * inv_cnt;
                    const float wb = wd * wd;
                    const float var_t = fmaxf(wa - wb, 0.0f);
                    const float wmag = (pm + qcl) * inv_cnt * ((float)b * inv_cnt) + wb;
                    const float err_t = (kErrF + 8.0f) * kU * wmag;
                    // loosest clamp (Lm, Hp): first-order bound with the exact counts, see stack_fast.hip
                    float var_l;
                    {
                        const float dL = (wi.Lp - wi.Lm) * (1.0f + 2.0f * kU), dH = (wi.Hp - wi.Hm) * (1.0f + 2.0f * kU);
                        const float ybar = c + wd;
                        const float slop = (float)(LY::ROUNDINGS + 8) * kU * 1.01f * __builtin_amdgcn_sqrtf(wmag) +
                                           4.0f * kU * fabsf(ybar) + 1.0e-30f;
                        const float gL = fmaxf(ybar - wi.Lp, 0.0f) + slop, gH = fmaxf(wi.Hm - ybar, 0.0f) + slop;
                        const float corr = (n_lo * (dL * (2.0f * gL + dL)) + n_hi * (dH * (2.0f * gH + dH))) * inv_cnt;
                        var_l = var_t + ((corr == corr) ? corr * 1.001f : 0.0f);
                    }
                    const bool shape_ok = jl <= jh;
                    wi.finish_round(var_t, err_t, var_l, err_t, eps_r, e_m, shape_ok, inner, bail);
                }
                s_min = wi.hull_lo;
                s_max = wi.hull_hi;
            }

            // ---- the reference's bound expressions (stack.go:408-409) at both ends of the interval ----
            const float tl0 = __fmul_rn(p.sig_lo, s_min), tl1 = __fmul_rn(p.sig_lo, s_max);
            const float th0 = __fmul_rn(p.sig_hi, s_min), th1 = __fmul_rn(p.sig_hi, s_max);
            const float la = __fsub_rn(median, tl0), lb = __fsub_rn(median, tl1);
            const float ha = __fadd_rn(median, th0), hb = __fadd_rn(median, th1);
            const float lo_min = fminf(la, lb), lo_max = fmaxf(la, lb);
            const float hi_min = fminf(ha, hb), hi_max = fmaxf(ha, hb);

            // ---- clips: certain below lo_min / above hi_max; a sample between the two ends of a bound
            // interval is undecidable ----
            const bool counting = active && !bail && lo_max == lo_max && hi_min == hi_min;
            const int a1 = walk_up<PW, false>(x, a, b, lo_min, counting, top);
            const int b1 = walk_down<PW, false>(x, b, a1, hi_max, counting);
            const int c1 = a1 - a, d1 = b - b1;
            const float next_lo = x[min(a1, top) * PW], next_hi = x[max(b1 - 1, 0) * PW];
            const bool amb = (a1 < b1) && (next_lo < lo_max || next_hi > hi_min);
            if (active) {
                bail |= !counting || amb || (lo_max > hi_min && (c1 + d1) > 0);
                if (bail) {
                    to_exact = true;
                    active = false;
                } else {
                    c_lo += c1;
                    c_hi += d1;
                    a = a1;
                    b = b1;
                    if ((c1 + d1) == 0 || (b - a) <= 1) {      // stack.go:427-430: the mean BEFORE this pass
                        res = m;
                        active = false;
                    }
                }
            }
        }

#ifdef NL_ROUND_STATS
        if (lane == 0) {
            const unsigned long long t3 = __builtin_readcyclecounter();
            NL_STAT(5, t1 - t0); NL_STAT(6, t2 - t1); NL_STAT(7, t3 - t2);
        }
#endif
        const bool rep = on && role == 0;
        if (rep && !to_exact) {
            p.out[pix] = res;
            c_lo_total += c_lo;
            c_hi_total += c_hi;
        }
        if (rep && to_exact && p.nrounds) p.nrounds[pix] = 0;       // (no decided rounds on record for the replay)
        const unsigned long long em = __ballot(rep && to_exact);
        if (em) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
            base = __shfl(base, 0, 64);
            const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
            if (rep && to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
        }
    }

#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_lo_total += __shfl_xor(c_lo_total, o, 64);
        c_hi_total += __shfl_xor(c_hi_total, o, 64);
    }
    if (lane == 0) {
        unsigned long long *slot = clip_slot(p, block);
        if (c_lo_total) atomicAdd(slot + 0, (unsigned long long)c_lo_total);
        if (c_hi_total) atomicAdd(slot + 1, (unsigned long long)c_hi_total);
    }
}

template <int LPP, bool WINSOR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
void stack_sigma_mlg_kernel(StackArgs p, FastArgs q)
{
    mlg_body<LPP, WINSOR>(p, q, blockIdx.x, gridDim.x);
}

#ifndef NL_TAIL_FUSED_TU          // (stack_tail_fused.hip includes this file for mlg_body only)
// generic pass over fargs.in_list (the hand-over list of a zonal kernel): 2 or 4 lanes per pixel for
// 129..512 frames; one lane per pixel (64 pixels per wave, the same 48 KiB) for the winsorized
// one-lane kernels of stack_fast.hip, whose register version of this pass runs 28 lock-step
// winsorization rounds over all 128 masked positions -- 1.0 ms for the 57 k border pixels of a
// 128 x 4096^2 stack, less than one wave per SIMD, pure latency
hipError_t launch_stack_sigma_mlg(const StackArgs &args, const FastArgs &fargs, unsigned grid, hipStream_t stream,
                                  bool winsor)
{
    if (args.n_frames <= kMlNS) {
        if (winsor) hipLaunchKernelGGL((stack_sigma_mlg_kernel<1, true>), dim3(grid), dim3(64), 0, stream, args, fargs);
        else        hipLaunchKernelGGL((stack_sigma_mlg_kernel<1, false>), dim3(grid), dim3(64), 0, stream, args, fargs);
    } else if (args.n_frames <= 2 * kMlNS) {
        if (winsor) hipLaunchKernelGGL((stack_sigma_mlg_kernel<2, true>), dim3(grid), dim3(64), 0, stream, args, fargs);
        else        hipLaunchKernelGGL((stack_sigma_mlg_kernel<2, false>), dim3(grid), dim3(64), 0, stream, args, fargs);
    } else {
        if (winsor) hipLaunchKernelGGL((stack_sigma_mlg_kernel<4, true>), dim3(grid), dim3(64), 0, stream, args, fargs);
        else        hipLaunchKernelGGL((stack_sigma_mlg_kernel<4, false>), dim3(grid), dim3(64), 0, stream, args, fargs);
    }
    return hipGetLastError();
}
#endif  // NL_TAIL_FUSED_TU

}  // namespace nl
