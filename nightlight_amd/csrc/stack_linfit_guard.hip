// stack_linfit_guard.hip -- GUARDED stages of StackLinearFit in front of the bit-exact cascade (round 5).
//
// Reference: internal/ops/stack/stack.go:834-918 (loop: sort, LinearRegression, mean absolute residual, reject,
// result = ymean of the last regression), internal/stats/stats.go:246-261, 569-586.
//
// The bit-exact kernel (stack_linfit.hip) replays four sequential fp32 sums per fit iteration; in chunks where some
// lane has dead samples that costs 45 instructions per sample and iteration.  A guarded stage decides the SAME
// rejects from quantities that do not depend on the reference's summation order:
//   * ymean is still formed exactly -- the reference's sequential sum in sorted order, one add per sample -- so the
//     result (stack.go:911) stays bit-identical;
//   * the reference's slope is  C_ref / (xstd * ystd * (n+1)) * ystd / xstd  (stats.go:576-581): its own ystddev
//     cancels up to five roundings, and C_ref is a sequential sum of the products fl(fl(i - xmean) * fl(y_i - ymean));
//     sum_i (i - xmean) = 0 exactly (xmean = (n-1)/2 is a half-integer), so in real arithmetic C = sum (i - xmean) z_i
//     for ANY shift z = y - c.  Hence  |slope_ref - C K| <= K gamma_(n+1) T + gamma_5 |slope|,  K = 1 / (xstd^2 (n+1)),
//     T = sum |i - xmean| |y_i - ymean|  (bounded below from the residuals: no second moment needed);
//   * residuals against the stage's own line (one fma, one subtraction per sample), their absolute sum, and the
//     reference's intercept / line / sigma / thresholds enclosed with the standard forward bounds (every term is
//     spelled out where it is computed).  A reject decision is taken only if the sample's distance from the
//     threshold exceeds the enclosure's half-width B; a pixel with ONE sample inside the band, a non-finite or
//     extreme fit (|y| >= 2^60, |slope| < 2^-60: the reference's intermediate products could leave the normal
//     range) is handed to the bit-exact cascade WITH its liveness mask and the counters so far -- it continues
//     there at the same iteration.
// tests/sweeps/linfit_guard_sim.py simulates exactly these formulas against the reference's fp32 arithmetic: on the
// bench distribution 6 % of the pixels are handed over (0 enclosure violations, 0 wrong decisions).
// Dead samples are +Inf in the registers (what the sort's padding already is): their residual is +Inf, so they
// never look undecidable; liveness bits mask them out of the sums.
#include <cstdio>
#include <cstdlib>
#include "linfit_common.hpp"

namespace nl {

#ifndef NL_LF_CHUNK
#define NL_LF_CHUNK 8
#endif

__device__ __forceinline__ float bfi_f(int mask, float a, float b)       // mask ? a : b, bitwise
{
    return __int_as_float((__float_as_int(a) & mask) | (__float_as_int(b) & ~mask));
}

// min(a, |b|, |c|) as ONE instruction (fabsf in front of an asm operand costs a v_and each)
__device__ __forceinline__ float min3_abs(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// x = mask ? +Inf : x, in place (a tied operand: the column keeps its registers across the branch around it)
__device__ __forceinline__ void kill_sample(float &x, int mask)
{
    asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x) : "v"(mask), "v"(0x7f800000));
}

template <int NS, bool CONT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS > 96 ? 3 : 1, 8)))
void stack_linfit_guard_kernel(StackArgs p, FastArgs q, LinfitStage g, LinfitGuardLists x)
{
    constexpr int NW = (NS + 31) / 32;          // liveness words per pixel
    const int lane = threadIdx.x & 63;
    int c_lo = 0, c_hi = 0;
    const int64_t limit = CONT ? (int64_t)min(*g.in_count, g.in_capacity) : p.npix;
    const int64_t sweep = CONT ? (int64_t)gridDim.x * blockDim.x : limit;
  for (int64_t wg_item = (int64_t)blockIdx.x * blockDim.x; wg_item < limit; wg_item += sweep) {
    int N = p.n_frames;
    asm volatile("" : "+s"(N));                 // per trip: keeps per-frame scalars out of the loop preheader
    const int64_t item = wg_item + threadIdx.x;
    const bool on = item < limit;
    int64_t pix = item;
    if (CONT) pix = on ? (int64_t)g.in_list[item] : 0;
    const unsigned boff = (unsigned)(on ? pix : 0) * 4u;
    float v[NS];
    const int n = gather_sorted<NS, 32>(p.frames, p.stride, N, boff, v);

    unsigned live[NW];
    static_range<0, NW>([&](auto W) NL_INL {
        constexpr int w = decltype(W)::value;
        const int c = min(max(n - 32 * w, 0), 32);
        live[w] = c >= 32 ? 0xFFFFFFFFu : ((1u << c) - 1u);
    });
    // a genuine +-Inf sample cannot be told from a pad / a dead sample afterwards -> exact kernel (as stack_linfit.hip);
    // amax = the largest magnitude among the pixel's samples (range guard of the enclosure)
    unsigned inf_any = 0;
    float amax = 0.0f;
    {
        int nn = n;
        static_chunks<0, NS, 8>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            if constexpr ((k & 7) == 0) nn = opaque(nn);
            const unsigned pad = (unsigned)((nn - 1 - k) >> 31);                // all ones for k >= n
            const unsigned mag = (unsigned)__float_as_int(v[k]) & 0x7fffffffu & ~pad;
            inf_any |= (mag == 0x7f800000u) ? 1u : 0u;
            amax = fmaxf(amax, __int_as_float((int)mag));
        });
    }
    const bool to_exact = inf_any != 0;
    // shift of the moment sums: a sample from the middle of the sorted column (any finite number would do; the
    // enclosure pays for a poor one through |ymean - c|)
    float cs = v[NS / 2];
    if (!(fabsf(cs) < __builtin_inff())) cs = v[0];
    int m_saved = n;
    if constexpr (CONT) {
        const uint4 st = g.in_state[on ? item : 0];
        const unsigned w4[4] = {st.x, st.y, st.z, st.w};
        m_saved = 0;
        static_range<0, NW>([&](auto W) NL_INL {
            constexpr int w = decltype(W)::value;
            live[w] = w4[w];
            m_saved += __popc(w4[w]);
        });
        // the samples the earlier stages rejected: dead = +Inf
        static_chunks<0, NS, 8>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            const int lm = (int)(live[k >> 5] << (31 - (k & 31))) >> 31;
            v[k] = bfi_f(lm, v[k], __builtin_inff());
        });
    }

    float res = p.ref_loc;
    int p_lo = 0, p_hi = 0;
    int m = m_saved;                            // surviving samples
    bool active = on && n > 0 && !to_exact;
    bool doubt = false;                         // handed to the bit-exact cascade (state = live, counters so far)
    int iters = 0;

    constexpr int CH = NL_LF_CHUNK, NC = NS / CH, CPW = 32 / CH;     // chunks per liveness word
    static_assert(NS % CH == 0, "chunks");
    while (__any(active) && (g.max_iters == 0 || iters < g.max_iters)) {
        iters++;
        const float fm = (float)m;
        const int mt = (active && m >= 1) ? m : 1;
        const float xm = p.xstat[2 * mt], xsd = p.xstat[2 * mt + 1];
        unsigned aa = 0, ad = 0;                // bit c: chunk c all alive / all dead in every fitting lane
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            constexpr unsigned full = (1u << CH) - 1u;
            const unsigned byte = (live[c / CPW] >> (CH * (c % CPW))) & full;
            aa |= (__ballot(active && byte != full) == 0ull ? 1u : 0u) << c;
            ad |= (__ballot(active && byte != 0u) == 0ull ? 1u : 0u) << c;
        });
        aa = (unsigned)__builtin_amdgcn_readfirstlane((int)aa);
        ad = (unsigned)__builtin_amdgcn_readfirstlane((int)ad);
#define NL_M(k) ((int)(live[(k) >> 5] << (31 - ((k) & 31))) >> 31)
#define NL_AND(x, m) __int_as_float(__float_as_int(x) & (m))
        // ---- pass A: the reference's sum of the ys (stats.go:248-251), sequential in sorted order = exact ymean;
        // beside it C = sum (rank - xmean) (y - c) in any order (four accumulators; <= 17 roundings on every path
        // from a term to the total: z, <= 8 inside a chunk, the chunk's fma, <= 4 + 2 between the chunks) ----
        float s = 0.0f, fi = 0.0f;
        float dxr = -xm;                        // rank - xmean of the next alive sample (exact: half-integers)
        float ca[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if ((aa >> c) & 1u) {
                NL_KEEP_BRANCH;
                float dc = 0.0f, pj = 0.0f;
                static_range<0, CH>([&](auto J) NL_INL {
                    constexpr int j = decltype(J)::value;
                    s = __fadd_rn(s, v[c * CH + j]);
                    const float z = __fsub_rn(v[c * CH + j], cs);
                    dc = j == 0 ? z : __fadd_rn(dc, z);
                    if constexpr (j == 1) pj = z;
                    if constexpr (j > 1) pj = __builtin_fmaf((float)j, z, pj);
                });
                ca[c & 3] = __fadd_rn(ca[c & 3], __builtin_fmaf(dxr, dc, pj));
                dxr += (float)CH;
            } else {
                NL_KEEP_BRANCH;
                static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const int lm = NL_M(k);
                    s = __fadd_rn(s, NL_AND(v[k], lm));
                    const float z = NL_AND(__fsub_rn(v[k], cs), lm);
                    ca[c & 3] = __builtin_fmaf(dxr, z, ca[c & 3]);
                    dxr += NL_AND(1.0f, lm);                             // index among the survivors
                });
            }
        });
        const float ym = s / fm;
        const float cz = __fadd_rn(__fadd_rn(ca[0], ca[1]), __fadd_rn(ca[2], ca[3]));
        // K = 1 / (xstd^2 (n+1)): hardware reciprocal (1 ulp) and three roundings -- inside the 16 u of A_s below
        const float kk = __builtin_amdgcn_rcpf(__fmul_rn(__fmul_rn(xsd, xsd), __fadd_rn(fm, 1.0f)));
        const float sl = __fmul_rn(cz, kk);
        const float icpt = __builtin_fmaf(-sl, xm, ym);
        // ---- pass B: residuals against the stage's own line, their absolute sum; extremes per 32 positions of the
        // all-alive chunks (they decide below which of those chunks the reject pass has to look at) ----
        float sa[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float dmax[NW], dmin[NW];
        static_range<0, NW>([&](auto W) NL_INL { dmax[decltype(W)::value] = -__builtin_inff(); dmin[decltype(W)::value] = __builtin_inff(); });
        fi = 0.0f;
        forget_words<NW>(live);
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if ((aa >> c) & 1u) {
                NL_KEEP_BRANCH;
                const float ic = __builtin_fmaf(fi, sl, icpt);
                float dprev = 0.0f;
                static_range<0, CH>([&](auto J) NL_INL {
                    constexpr int j = decltype(J)::value;
                    const float lin = j == 0 ? ic : __builtin_fmaf((float)j, sl, ic);
                    const float diff = __fsub_rn(v[c * CH + j], lin);
                    sa[j & 3] = __fadd_rn(sa[j & 3], fabsf(diff));
                    if constexpr ((j & 1) == 0) {
                        dprev = diff;
                    } else {
                        dmax[c / CPW] = max3_asm(dmax[c / CPW], dprev, diff);
                        dmin[c / CPW] = min3_asm(dmin[c / CPW], dprev, diff);
                    }
                });
                fi += (float)CH;
            } else {
                NL_KEEP_BRANCH;
                static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const int lm = NL_M(k);
                    const float lin = __builtin_fmaf(fi, sl, icpt);
                    const float diff = __fsub_rn(v[k], lin);
                    sa[k & 3] = __fadd_rn(sa[k & 3], fabsf(NL_AND(diff, lm)));
                    fi += NL_AND(1.0f, lm);
                });
            }
        });
        const float sabs = __fadd_rn(__fadd_rn(sa[0], sa[1]), __fadd_rn(sa[2], sa[3]));
        const float sg = sabs / fm;
        const float lb = __fmul_rn(p.sig_lo, sg), hb = __fmul_rn(p.sig_hi, sg);
        // ---- the enclosure (every factor 1.00x absorbs the roundings of the bound's own arithmetic) ----
        // T >= sum |i - xm| |y_i - ymean|:  |y_i - ymean| <= |r_i| + |slope| |i - xm| + 3 u Lmax  (r_i, slope: this stage's)
        const float as = fabsf(sl), m1 = fm - 1.0f;
        const float xvar = (fm * fm - 1.0f) * (1.0f / 12.0f);
        const float lmax0 = fabsf(ym) + as * m1 * 0.51f;
        const float tt = 0.5f * m1 * sabs + as * fm * xvar + (3.0f * kU) * lmax0 * fm * fm * 0.25f;
        // the reference's correlation sum: two roundings per term, n - 1 additions; this stage's own: 17 roundings,
        // over sum |i - xm| |y_i - c| <= T + |ymean - c| n^2 / 4
        const float ac = ((fm + 2.0f) * kU * tt + (17.0f * kU) * (tt + fabsf(ym - cs) * fm * fm * 0.25f)) * 1.002f + 1e-30f;
        const float a_s = kk * ac * 1.001f + (16.0f * kU) * as;        // |slope_ref - sl|
        const float lmax = fabsf(ym) + (as + a_s) * m1 * 0.5f;          // >= |line| of either fit at every rank
        // |lin_ref(i) - lin(i)| <= a_s |i - xm| + E: four roundings of the reference's line (slope*xm, the intercept,
        // i*slope, the sum), three of this stage's (the intercept, the chunk's base, the sample's fma)
        const float ee = kU * (1.5f * (as + a_s) * m1 + 5.0f * lmax) * 1.01f;
        const float ww = a_s * m1 * 0.5f + ee;
        // |sigma_ref - sg|: mean |lin_ref - lin| (mean |i - xm| <= n / 4), the reference's sequential sum (gamma_n), this
        // stage's (four accumulators), the divisions
        const float asg = (a_s * fm * 0.25f + ee) + (1.25f * fm + 12.0f) * kU * 1.001f * sg;
        const float bl = ww + fabsf(p.sig_lo) * asg + (4.0f * kU) * fabsf(lb);
        const float bh = ww + fabsf(p.sig_hi) * asg + (4.0f * kU) * fabsf(hb);
        const float band = fmaxf(bl, bh) * 1.01f + 1e-30f;
        const bool ok = (band < __builtin_inff()) && (fabsf(ym) < __builtin_inff()) && as > 8.673617e-19f /* 2^-60 */ &&
                        amax < 1.1529215e18f /* 2^60 */ && m >= 2;
        // all-alive chunks some of whose samples may be rejected (or undecidable) in some lane: per 32 positions
        unsigned holed = 0;
        static_range<0, NW>([&](auto W) NL_INL {
            constexpr int w = decltype(W)::value;
            holed |= (__any(active && ok && (dmin[w] < band - lb || dmax[w] > hb - band)) ? 1u : 0u) << w;
        });
        holed = (unsigned)__builtin_amdgcn_readfirstlane((int)holed);
        // ---- pass C: reject decisions (stack.go:890-904) with the band ----
        // reference: lin - g > lb -> low, else g - lin > hb -> high.  r + lb < 0 <=> low; hb - r < 0 <=> high (the
        // reference's two differences are each other's negatives); decided only if no alive sample has
        // min(|r + lb|, |hb - r|) <= band.
        unsigned lo_n = 0, hi_n = 0;
        unsigned nlive[NW];
        static_range<0, NW>([&](auto W) NL_INL { nlive[decltype(W)::value] = live[decltype(W)::value]; });
        float amin = __builtin_inff();
        fi = 0.0f;
        forget_words<NW>(live);
        const float sl2 = opaque_f(sl);
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if (((aa >> c) & 1u) && !((holed >> (c / CPW)) & 1u)) {
                NL_KEEP_BRANCH;
                fi += (float)CH;
            } else {
                NL_KEEP_BRANCH;
                unsigned lowb = 0, highb = 0;
                static_range<0, CH>([&](auto J) NL_INL {
                    constexpr int k = c * CH + decltype(J)::value;
                    const float lin = __builtin_fmaf(fi, sl2, icpt);
                    const float r = __fsub_rn(v[k], lin);
                    const float alo = __fadd_rn(r, lb), ahi = __fsub_rn(hb, r);
                    lowb = __builtin_amdgcn_alignbit(lowb, (unsigned)__float_as_int(alo), 31);
                    highb = __builtin_amdgcn_alignbit(highb, (unsigned)__float_as_int(ahi), 31);
                    amin = min3_abs(amin, alo, ahi);
                    fi += NL_AND(1.0f, NL_M(k));
                });
                constexpr int sh = CH * (c % CPW);
                constexpr unsigned full = (1u << CH) - 1u;
                const unsigned alive = (live[c / CPW] >> sh) & full;
                const unsigned low = (__builtin_bitreverse32(lowb) >> (32 - CH)) & alive;       // sample j of the chunk at bit j
                const unsigned high = (__builtin_bitreverse32(highb) >> (32 - CH)) & alive & ~low;
                const unsigned gone = low | high;
                lo_n = opaque_u(lo_n + (unsigned)__popc(low));
                hi_n = opaque_u(hi_n + (unsigned)__popc(high));
                nlive[c / CPW] = opaque_u(nlive[c / CPW] & ~(gone << sh));
                if (__any(active && gone != 0u)) {      // the rejected samples become dead: +Inf in the column
                    NL_KEEP_BRANCH;
                    static_range<0, CH>([&](auto J) NL_INL {
                        constexpr int j = decltype(J)::value;
                        kill_sample(v[c * CH + j], (int)(gone << (31 - j)) >> 31);
                    });
                }
            }
        });
#undef NL_M
#undef NL_AND
        if (active) {
            if (!ok || !(amin > band)) {
                doubt = true;                                   // this iteration is the bit-exact cascade's
                active = false;
            } else {
                p_lo += (int)lo_n;
                p_hi += (int)hi_n;
                const int left = (int)(lo_n + hi_n);
                res = ym;                                       // stack.go:911: mean of the last regression
                if (left == 0 || m < 3) active = false;
                m -= left;
                static_range<0, NW>([&](auto W) NL_INL { live[decltype(W)::value] = nlive[decltype(W)::value]; });
            }
        }
    }

    // lanes that are still fitting after this stage's quota go to the next guarded stage, undecidable ones to the
    // bit-exact cascade; the rejections made so far are final and are counted here
    const bool more = active;
    if (on && !to_exact && !more && !doubt) p.out[pix] = res;
    if (on && !to_exact) { c_lo += p_lo; c_hi += p_hi; }
    unsigned w4[4] = {0u, 0u, 0u, 0u};
    static_range<0, NW>([&](auto W) NL_INL { w4[decltype(W)::value] = live[decltype(W)::value]; });
    const unsigned long long mm = __ballot(more);
    if (mm) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(g.out_count, (unsigned)__popcll(mm));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(mm & ((1ull << lane) - 1ull));
        if (more && slot < g.out_capacity) {
            g.out_list[slot] = (unsigned)pix;
            g.out_state[slot] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    const unsigned long long dm = __ballot(doubt);
    if (dm) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(x.x_count, (unsigned)__popcll(dm));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(dm & ((1ull << lane) - 1ull));
        if (doubt && slot < x.x_capacity) {
            x.x_list[slot] = (unsigned)pix;
            x.x_state[slot] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    const unsigned long long em = __ballot(on && to_exact);
    if (em) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
        if (on && to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
    }
  }

    __shared__ int s_lo[4], s_hi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_lo += __shfl_xor(c_lo, o, 64);
        c_hi += __shfl_xor(c_hi, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = c_lo; s_hi[threadIdx.x >> 6] = c_hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t_lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
        const int t_hi = s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3];
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}

// ---- cascade: guarded stages G0 (the tile), G1, G2 over the continuation lists; then the bit-exact stages of
// stack_linfit.hip over the pixels handed over (and what the first of them leaves) ----
// counters (LinfitGuardBufs::count): 0 = G0's continuation list, 1 = G1's, 2 = the hand-over list X (all guarded
// stages append), 3 = what the first bit-exact stage over X leaves
static const int *guard_quota()
{
    static int quota[2] = {8, 8};
    static bool parsed = false;
    if (!parsed) {
        parsed = true;
        if (const char *e = getenv("NL_LFG_QUOTA")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { quota[0] = a; quota[1] = b; }
        }
    }
    return quota;
}

template <int NS>
static void launch_lfg(const StackArgs &args, const FastArgs &f, const LinfitGuardBufs &b, unsigned blocks,
                       hipStream_t stream, hipEvent_t dominant_done)
{
    const int *quota = guard_quota();
    LinfitGuardLists xl = {b.list[2], b.count + 2, b.state[2], b.capacity};
    for (int s = 0; s < 3; s++) {
        LinfitStage g = {};
        g.max_iters = s < 2 ? quota[s] : 0;
        g.in_list = s ? b.list[s - 1] : nullptr;
        g.in_state = s ? b.state[s - 1] : nullptr;
        g.in_count = s ? b.count + (s - 1) : nullptr;
        g.in_capacity = b.capacity;
        g.out_list = b.list[s & 1];
        g.out_state = b.state[s & 1];
        g.out_count = b.count + (s < 2 ? s : 4);             // (the last stage runs to the end: nothing is appended)
        g.out_capacity = b.capacity;
        if (s == 0) {
            hipLaunchKernelGGL((stack_linfit_guard_kernel<NS, false>), dim3(blocks), dim3(256), 0, stream, args, f, g, xl);
            if (dominant_done) (void)hipEventRecord(dominant_done, stream);
        } else {
            const unsigned gblocks = blocks < 8192u ? blocks : 8192u;
            hipLaunchKernelGGL((stack_linfit_guard_kernel<NS, true>), dim3(gblocks), dim3(256), 0, stream, args, f, g, xl);
        }
    }
    // the pixels the guarded stages could not decide: bit-exact, from the state they were handed over with
    LinfitStage e0 = {};
    e0.max_iters = 8;
    e0.in_list = b.list[2]; e0.in_state = b.state[2]; e0.in_count = b.count + 2; e0.in_capacity = b.capacity;
    e0.out_list = b.list[0]; e0.out_state = b.state[0]; e0.out_count = b.count + 3; e0.out_capacity = b.capacity;
    launch_linfit_exact_stage(args, f, e0, blocks < 4096u ? blocks : 4096u, stream);
    LinfitStage e1 = {};
    e1.max_iters = 0;
    e1.in_list = b.list[0]; e1.in_state = b.state[0]; e1.in_count = b.count + 3; e1.in_capacity = b.capacity;
    e1.out_list = b.list[1]; e1.out_state = b.state[1]; e1.out_count = b.count + 5; e1.out_capacity = b.capacity;
    launch_linfit_exact_stage(args, f, e1, blocks < 2048u ? blocks : 2048u, stream);
}

int linfit_guard_supported(int mode, int n_frames, int64_t npix)
{
    return (mode == NL_ST_LINEAR_FIT && n_frames > 16 && n_frames <= 128 && npix < kFastMaxPixels) ? 1 : 0;
}

hipError_t launch_stack_linfit_guarded(const StackArgs &args, const FastArgs &fargs, const LinfitGuardBufs &bufs,
                                       hipStream_t stream, const char **name, hipEvent_t dominant_done)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    if (n <= 32)       { *name = "stack_linfit_guard_kernel<32, false>";  launch_lfg<32>(args, fargs, bufs, blocks, stream, dominant_done); }
    else if (n <= 48)  { *name = "stack_linfit_guard_kernel<48, false>";  launch_lfg<48>(args, fargs, bufs, blocks, stream, dominant_done); }
    else if (n <= 64)  { *name = "stack_linfit_guard_kernel<64, false>";  launch_lfg<64>(args, fargs, bufs, blocks, stream, dominant_done); }
    else if (n <= 96)  { *name = "stack_linfit_guard_kernel<96, false>";  launch_lfg<96>(args, fargs, bufs, blocks, stream, dominant_done); }
    else               { *name = "stack_linfit_guard_kernel<128, false>"; launch_lfg<128>(args, fargs, bufs, blocks, stream, dominant_done); }
    return hipGetLastError();
}

}  // namespace nl
