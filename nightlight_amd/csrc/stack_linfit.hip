// stack_linfit.hip -- register-resident StackLinearFit for gfx950, BIT-EXACT.
//
// Reference: internal/ops/stack/stack.go:834-918, internal/stats/stats.go:569-586.
// Every iteration of the reference sorts the surviving samples and then sums
// over them IN SORTED ORDER (MeanStdDev of the ys, the correlation sum, the
// mean absolute deviation).  A sorted sequence is unique, and removing the
// rejected samples from a sorted sequence leaves it sorted, so:
//   * one sorting network at the start replaces every QSortFloat32 call;
//   * the rejected samples are just marked dead in a per-pixel bit mask and
//     skipped (their term is multiplied by 0.0f, which leaves the sum unchanged);
//   * each sum is accumulated sequentially in register order = sorted order,
//     with the reference's own fp32 operations (never fused), so every
//     intermediate -- slope, intercept, sigma, the reject decisions, the
//     counters and the result -- is bit-identical to the reference.
// One pixel per lane, samples in VGPRs, no LDS, no hand-over lists (a pixel
// with a +-Inf sample is replayed by the LDS kernel: its pads are +Inf too).
// MeanStdDev of xs = 0..m-1 depends on m only and comes from the host table.
#include <cstdio>
#include <cstdlib>
#include "linfit_common.hpp"

namespace nl {

#ifndef NL_LF_CHUNK
#define NL_LF_CHUNK 8
#endif
// Parked chunks (round 6, NS = 128 only).  At three waves per SIMD the kernel has 168 registers, 128 of them the column: the
// allocator spilled 9 - 31 of the rest to scratch and re-loaded ~20 of them per fit iteration -- SQ_WAIT_ANY (waves parked at
// s_waitcnt) was 25 % of the wave cycles of the first stage (profiles/r06_linfit_pmc.txt), and three waves do not hide it.  The
// NL_LF_PARK lowest and NL_LF_PARK highest chunks of the sorted column therefore live in LDS instead ([slot][thread]:
// ds_read2st64_b32 with constant offsets, conflict-free) and are read back chunk by chunk inside the sweeps.  The ends are the
// chunks that die first (the reference rejects from the ends inwards; pads sit on top), so their reads fade out with the
// iterations.  Measured on 128 / 112 / 100 frames x 4096^2 (profiles/r06_linfit_park.txt): nothing parked 16.8 / 14.8 / 13.5 ms,
// one chunk per end 16.0 / 13.9 / 12.7, two 16.2 / 14.0 / 12.8, three 16.8 / 14.3 / 12.9; every fourth chunk with its reads issued
// one chunk ahead 16.1 - 16.4 / 14.1 (the reads ahead cost the registers the parking freed).  A fourth wave is out of reach: 128
// registers would leave ~70 for the column, i.e. 58 parked samples = 14.5 KiB per wave, and 16 waves x 14.5 KiB exceed the
// CU's 160 KiB.  NL_LF_PARK=0: everything in registers (the kernel of rounds 2 - 5).
#ifndef NL_LF_PARK
#define NL_LF_PARK 1
#endif
template <int NS, int CH>
struct LfPark {
    static constexpr int NC = NS / CH;
    static constexpr int ends = (NS == 128) ? NL_LF_PARK : 0;
    static constexpr int slots = 2 * ends * CH;                       // parked samples per pixel
    static constexpr __host__ __device__ bool parked(int c) { return c < ends || c >= NC - ends; }
    static constexpr __host__ __device__ int slot(int c) { return c < ends ? c : c - (NC - 2 * ends); }
};

template <int NS, bool CONT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS > 96 ? 3 : 1, 8)))
void stack_linfit_fast_kernel(StackArgs p, FastArgs q, LinfitStage g)
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

    // The sorted column never changes; a sample's liveness is one bit of
    // live[].  Initially the n valid samples (positions 0..n-1) are alive.
    // Pads are +Inf: replaced by 0 so that dead positions stay finite (they are
    // multiplied by 0.0f below).  A genuine +-Inf sample cannot be told from a
    // pad afterwards -> exact kernel.
    unsigned live[NW];
    static_range<0, NW>([&](auto W) NL_INL {
        constexpr int w = decltype(W)::value;
        const int c = min(max(n - 32 * w, 0), 32);
        live[w] = c >= 32 ? 0xFFFFFFFFu : ((1u << c) - 1u);
    });
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
    }
    unsigned inf_any = 0;
    {
        int nn = n;
        static_chunks<0, NS, 8>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            if constexpr ((k & 7) == 0) nn = opaque(nn);
            const unsigned pad = (unsigned)((nn - 1 - k) >> 31);                // all ones for k >= n
            const unsigned bits = (unsigned)__float_as_int(v[k]);
            inf_any |= (((bits & 0x7fffffffu) == 0x7f800000u) ? 1u : 0u) & ~pad;
            v[k] = __int_as_float((int)(bits & ~pad));                       // pads -> +0.0f
        });
    }
    const bool to_exact = inf_any != 0;

    using Park = LfPark<NS, NL_LF_CHUNK>;
    __shared__ float park[Park::slots > 0 ? Park::slots : 1][256];
    if constexpr (Park::slots > 0) {
        static_range<0, NS / NL_LF_CHUNK>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if constexpr (Park::parked(c))
                static_range<0, NL_LF_CHUNK>([&](auto J) NL_INL {
                    constexpr int j = decltype(J)::value;
                    park[Park::slot(c) * NL_LF_CHUNK + j][threadIdx.x] = v[c * NL_LF_CHUNK + j];
                });
        });
    }
    // x[0..CH) = the samples of chunk c: registers, or (parked chunks) this thread's column slots in LDS
#define NL_CHUNK_VALS(c, x)                                                                                  \
    float x[NL_LF_CHUNK];                                                                                    \
    static_range<0, NL_LF_CHUNK>([&](auto J_) NL_INL {                                                        \
        constexpr int j_ = decltype(J_)::value;                                                              \
        if constexpr (Park::slots > 0 && Park::parked(c)) x[j_] = park[Park::slot(c) * NL_LF_CHUNK + j_][threadIdx.x]; \
        else x[j_] = v[(c) * NL_LF_CHUNK + j_];                                                              \
    })

    float res = p.ref_loc;
    int p_lo = 0, p_hi = 0;
    int m = m_saved;                            // surviving samples
    bool active = on && n > 0 && !to_exact;
    int iters = 0;

    // Chunks of 8 positions, classified per iteration for the whole wave (scalar masks): in a
    // chunk where every fitting lane has all 8 samples alive the sums need no liveness
    // arithmetic (x * 1.0f == x: the same bits) and the index among the survivors is the
    // chunk's first index plus a constant -- 14 instead of 45 instructions per sample and
    // iteration; a chunk that is dead in every fitting lane (pads, the ends of the column in
    // late iterations) is skipped; only mixed chunks -- where the rejects are happening --
    // run the masked code.  The reference rejects from the ends of the sorted column inwards
    // (12 + 12 of 128 samples on the bench stack) and, in late iterations, singles in the
    // middle; 10-11 of 16 chunks are fully alive on average.
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
        // liveness of position k as an all-ones / zero word: x & m is x or +0 -- a dead sample adds
        // +0 to a sum, as skipping it does (an accumulator that starts at +0 never becomes -0).
        // (Not x * 0.0f: that is NaN for a dead sample whose square overflowed, and the compiler
        // pairs such multiplies into v_pk_mul_f32, which wants a second, pair-aligned copy of the column.)
#define NL_M(k) ((int)(live[(k) >> 5] << (31 - ((k) & 31))) >> 31)
#define NL_AND(x, m) __int_as_float(__float_as_int(x) & (m))
        // ---- MeanStdDev(ys), stats.go:246-261, sequential in sorted order ----
        float s = 0.0f;
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if ((aa >> c) & 1u) {
                NL_KEEP_BRANCH;
                NL_CHUNK_VALS(c, x);
                static_range<0, CH>([&](auto J) NL_INL { s = __fadd_rn(s, x[decltype(J)::value]); });
            } else {
                NL_KEEP_BRANCH;
                NL_CHUNK_VALS(c, x);
                static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    s = __fadd_rn(s, NL_AND(x[k - c * CH], NL_M(k)));
                });
            }
        });
        const float ym = s / fm;
        // ---- variance of the ys and the correlation sum (stats.go:573-579; divisor n+1,
        // quirk Q5) in one pass: both only need ymean, and each accumulator still sees its
        // terms in index order ----
        float vs = 0.0f, corr = 0.0f, fi = 0.0f;
        forget_words<NW>(live);
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if ((aa >> c) & 1u) {
                NL_KEEP_BRANCH;
                // fl(i - xm) is exact (half-integers below 2^23), so (fi - xm) + j is the same float
                const float dxb = __fsub_rn(fi, xm);
                NL_CHUNK_VALS(c, x);
                static_range<0, CH>([&](auto J) NL_INL {
                    constexpr int j = decltype(J)::value;
                    const float dy = __fsub_rn(x[j], ym);
                    vs = __fadd_rn(vs, __fmul_rn(dy, dy));
                    corr = __fadd_rn(corr, __fmul_rn(__fadd_rn(dxb, (float)j), dy));
                });
                fi += (float)CH;
            } else {
                NL_KEEP_BRANCH;
                NL_CHUNK_VALS(c, x);
                static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const int lm = NL_M(k);
                    const float dy = __fsub_rn(x[k - c * CH], ym);
                    const float dd = __fmul_rn(dy, dy);
                    vs = __fadd_rn(vs, NL_AND(dd, lm));
                    const float dx = __fsub_rn(fi, xm);
                    const float t = __fmul_rn(dx, dy);
                    corr = __fadd_rn(corr, NL_AND(t, lm));
                    fi += NL_AND(1.0f, lm);                              // index among the survivors
                });
            }
        });
        const float ysd = sqrt_go(vs / fm);
        float den = __fmul_rn(xsd, ysd);
        den = __fmul_rn(den, __fadd_rn(fm, 1.0f));
        corr = corr / den;
        float slope = __fmul_rn(corr, ysd);
        slope = slope / xsd;
        float icpt = __fsub_rn(ym, __fmul_rn(slope, xm));
        // ---- mean absolute deviation from the fit, stack.go:879-886 ----
        // (a NaN fit -- ystddev 0 -- makes every term NaN in the reference too)
        // In the all-alive chunks the extremes of the residuals are kept: they decide below
        // whether any of those samples is rejected (fl(lin - g) == -fl(g - lin), so the one
        // difference serves both of the reference's tests).  max / min as the instructions, two
        // residuals each: the compiler re-associates fmaxf / fminf chains into a tree over all
        // the residuals and spills the column for it.
        float sg = 0.0f;
        // (per 32 positions.  Per chunk, for the shallow columns whose iterations are issue-bound: 25 / 32 / 48 / 64 frames
        // 2.35 / 2.68 / 4.30 / 6.03 -> 2.42 / 2.70 / 4.25 / 6.04 ms, round 6 -- the masked reject pass over chunks without a
        // candidate is not where a shallow iteration's time goes)
        float dmax[NW], dmin[NW];
        static_range<0, NW>([&](auto W) NL_INL { dmax[decltype(W)::value] = -__builtin_inff(); dmin[decltype(W)::value] = __builtin_inff(); });
        fi = 0.0f;
        forget_words<NW>(live);
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if ((aa >> c) & 1u) {
                NL_KEEP_BRANCH;
                float dprev = 0.0f;
                NL_CHUNK_VALS(c, x);
                static_range<0, CH>([&](auto J) NL_INL {
                    constexpr int j = decltype(J)::value;
                    const float lin = __fadd_rn(__fmul_rn(__fadd_rn(fi, (float)j), slope), icpt);
                    const float diff = __fsub_rn(x[j], lin);
                    sg = __fadd_rn(sg, fabsf(diff));
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
                NL_CHUNK_VALS(c, x);
                static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                    constexpr int k = decltype(K)::value;
                    const int lm = NL_M(k);
                    const float lin = __fadd_rn(__fmul_rn(fi, slope), icpt);
                    const float diff = __fsub_rn(x[k - c * CH], lin);
                    sg = __fadd_rn(sg, NL_AND(fabsf(diff), lm));
                    fi += NL_AND(1.0f, lm);
                });
            }
        });
        sg = sg / fm;
        // ---- reject, stack.go:890-904 ----
        // Reference: lin-g > lb -> low, else g-lin > hb -> high; any NaN in the fit
        // makes both comparisons false.  Done with sign bits: lb - (lin-g) < 0 <=>
        // lin-g > lb.  A NaN fit is neutralised first (flat line, infinite bounds).
        float lb = __fmul_rn(p.sig_lo, sg), hb = __fmul_rn(p.sig_hi, sg);
        const bool bad = !(slope == slope) || !(icpt == icpt) || !(lb == lb) || !(hb == hb);
        if (bad) { slope = 0.0f; icpt = 0.0f; lb = __builtin_inff(); hb = __builtin_inff(); }
        // a sample of an all-alive chunk is rejected in some lane (lin - g > lb <=> diff < -lb,
        // g - lin > hb <=> diff > hb; a NaN fit rejects nothing): then those chunks take the masked
        // pass as well -- their bits are all set, and it finds nothing in the other lanes.  Decided per
        // 32 positions: it happens in two of three iterations of a wave (some lane's clipped end
        // reaches the next chunk, or a late fit rejects a single in the middle).
        unsigned holed = 0;
        static_range<0, NW>([&](auto W) NL_INL {
            constexpr int w = decltype(W)::value;
            holed |= (__any(active && !bad && (dmin[w] < -lb || dmax[w] > hb)) ? 1u : 0u) << w;
        });
        holed = (unsigned)__builtin_amdgcn_readfirstlane((int)holed);
        unsigned lo_n = 0, hi_n = 0;
        unsigned nlive[NW];
        static_range<0, NW>([&](auto W) NL_INL { nlive[decltype(W)::value] = live[decltype(W)::value]; });
        fi = 0.0f;
        forget_words<NW>(live);
        slope = opaque_f(slope);
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if (((aa >> c) & 1u) && !((holed >> (c / CPW)) & 1u)) {
                NL_KEEP_BRANCH;
                fi += (float)CH;
            } else {
                NL_KEEP_BRANCH;
                // sign bits of lb - (lin - g) and hb - (g - lin) = hb + (lin - g) (the same float:
                // fl(g - lin) == -fl(lin - g)), shifted into one word per test with v_alignbit;
                // liveness, counts and the mask update once per chunk
                unsigned lowb = 0, highb = 0;
                NL_CHUNK_VALS(c, x);
                static_range<0, CH>([&](auto J) NL_INL {
                    constexpr int k = c * CH + decltype(J)::value;
                    const float lin = __fadd_rn(__fmul_rn(fi, slope), icpt);
                    const float t = __fsub_rn(lin, x[k - c * CH]);
                    lowb = __builtin_amdgcn_alignbit(lowb, (unsigned)__float_as_int(__fsub_rn(lb, t)), 31);
                    highb = __builtin_amdgcn_alignbit(highb, (unsigned)__float_as_int(__fadd_rn(hb, t)), 31);
                    fi += NL_AND(1.0f, NL_M(k));
                });
                constexpr int sh = CH * (c % CPW);
                constexpr unsigned full = (1u << CH) - 1u;
                const unsigned alive = (live[c / CPW] >> sh) & full;
                const unsigned low = (__builtin_bitreverse32(lowb) >> (32 - CH)) & alive;       // sample j of the chunk at bit j
                const unsigned high = (__builtin_bitreverse32(highb) >> (32 - CH)) & alive & ~low;
                lo_n = opaque_u(lo_n + (unsigned)__popc(low));
                hi_n = opaque_u(hi_n + (unsigned)__popc(high));
                nlive[c / CPW] = opaque_u(nlive[c / CPW] & ~((low | high) << sh));
            }
        });
#undef NL_M
#undef NL_AND
        if (active) {
            p_lo += (int)lo_n;
            p_hi += (int)hi_n;
            const int left = (int)(lo_n + hi_n);
            res = ym;                                       // stack.go:911: mean of the last regression
            if (left == 0 || m < 3) active = false;
            m -= left;
            static_range<0, NW>([&](auto W) NL_INL { live[decltype(W)::value] = nlive[decltype(W)::value]; });
        }
    }

#undef NL_CHUNK_VALS

    // lanes that are still fitting after this stage's quota go to the next stage; the
    // rejections they made so far are final and are counted here
    const bool more = active;
    if (on && !to_exact && !more) NL_STORE_RESULT(&p.out[pix], res);
    if (on && !to_exact) { c_lo += p_lo; c_hi += p_hi; }
    const unsigned long long mm = __ballot(more);
    if (mm) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(g.out_count, (unsigned)__popcll(mm));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(mm & ((1ull << lane) - 1ull));
        if (more && slot < g.out_capacity) {
            g.out_list[slot] = (unsigned)pix;
            unsigned w4[4] = {0u, 0u, 0u, 0u};
            static_range<0, NW>([&](auto W) NL_INL { w4[decltype(W)::value] = live[decltype(W)::value]; });
            g.out_state[slot] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
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

// ---- 129 .. 512 frames: 2 or 4 lanes per pixel ---------------------------------------
// After ml_gather_sorted lane r of a pixel holds the sorted ranks [128r, 128r+128).  The
// reference's sums run over the sorted samples in index order, i.e. lane 0's column, then
// lane 1's, ...: a chain that crosses the lanes.  Every chained pass is therefore issued LPP
// times; in round j the lanes start from the value lane j-1 finished with (broadcast inside
// the quad) and only lane j's result is kept.  The index among the survivors at the start
// of a lane, the reject flags and the counters do not depend on the chain and are formed in
// parallel.  Same fp32 operations in the same order as the one-lane kernel: bit-exact.
template <int LPP, int J>
__device__ __forceinline__ float quad_from(float x)       // value of the pixel's lane J, in every lane
{
    if constexpr (LPP == 2) return dpp_f<J == 0 ? 0xA0 : 0xF5>(x);            // quad_perm [0,0,2,2] / [1,1,3,3]
    else return dpp_f<J == 0 ? 0x00 : (J == 1 ? 0x55 : (J == 2 ? 0xAA : 0xFF))>(x);
}

template <int LPP, bool CONT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void stack_linfit_ml_kernel(StackArgs p, FastArgs q, LinfitStage g)
{
    constexpr int NS = kMlNS, NW = NS / 32;
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x % LPP;
    int c_lo = 0, c_hi = 0;
    const int64_t limit = CONT ? (int64_t)min(*g.in_count, g.in_capacity) : p.npix;
    const int64_t per_wg = blockDim.x / LPP;
    const int64_t sweep = CONT ? (int64_t)gridDim.x * per_wg : limit;
  for (int64_t wg_item = (int64_t)blockIdx.x * per_wg; wg_item < limit; wg_item += sweep) {
    const int64_t item = wg_item + threadIdx.x / LPP;
    const bool on = item < limit;
    int64_t pix = item;
    if (CONT) pix = on ? (int64_t)g.in_list[item] : 0;
    int N = p.n_frames;
    asm volatile("" : "+s"(N));
    float v[NS];
    const int n = ml_gather_sorted<LPP, NS, false>(p.frames, p.stride, N, on, pix, role, v);
    const int n_loc = min(max(n - role * NS, 0), NS);           // valid samples in this lane

    unsigned live[NW];
    static_range<0, NW>([&](auto W) NL_INL {
        constexpr int w = decltype(W)::value;
        const int c = min(max(n_loc - 32 * w, 0), 32);
        live[w] = c >= 32 ? 0xFFFFFFFFu : ((1u << c) - 1u);
    });
    int m_saved = n;
    if constexpr (CONT) {               // liveness masks of this lane's ranks, saved by the previous stage
        const uint4 st = g.in_state[(on ? item : 0) * LPP + role];
        live[0] = st.x; live[1] = st.y; live[2] = st.z; live[3] = st.w;
        m_saved = quad_sum<LPP>((int)(__popc(st.x) + __popc(st.y) + __popc(st.z) + __popc(st.w)));
    }
    unsigned inf_loc = 0;
    {
        int nn = n_loc;
        static_chunks<0, NS, 8>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            if constexpr ((k & 7) == 0) nn = opaque(nn);
            const unsigned pad = (unsigned)((nn - 1 - k) >> 31);
            const unsigned bits = (unsigned)__float_as_int(v[k]);
            inf_loc |= (((bits & 0x7fffffffu) == 0x7f800000u) ? 1u : 0u) & ~pad;
            v[k] = __int_as_float((int)(bits & ~pad));                       // pads -> +0.0f
        });
    }
    const bool to_exact = quad_or<LPP>((int)inf_loc) != 0;

    float res = p.ref_loc;
    int p_lo = 0, p_hi = 0;
    int m = m_saved;
    bool active = on && n > 0 && !to_exact;
    int iters = 0;

    // chunk classes as in the one-lane kernel: a chunk index is classified over ALL lanes of the wave
    // (lane r of a pixel holds ranks [128 r, 128 r + 128), so the clipped ends of a column sit in the
    // low chunks of its first and the high chunks of its last lane)
    constexpr int CH = NL_LF_CHUNK, NC = NS / CH, CPW = 32 / CH;
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
        // survivors in the lanes before this one = index among the survivors of this lane's first
        int live_loc = 0;
        static_range<0, NW>([&](auto W) NL_INL { live_loc += __popc(live[decltype(W)::value]); });
        int before = 0;
        {
            const int l1 = dpp_i<kSwap1>(live_loc);
            if constexpr (LPP == 2) {
                before = (role & 1) ? l1 : 0;
            } else {
                const int pair = live_loc + l1;                       // lanes {0,1} or {2,3}
                const int other = dpp_i<kSwap2>(pair);
                before = ((role & 1) ? l1 : 0) + ((role & 2) ? other : 0);
            }
        }
        const float fi0 = (float)before;
#define NL_M(k) ((int)(live[(k) >> 5] << (31 - ((k) & 31))) >> 31)
#define NL_AND(x, m) __int_as_float(__float_as_int(x) & (m))
        // ---- sum of the ys (stats.go:248-251), chained through the lanes ----
        float s = 0.0f;
        static_range<0, LPP>([&](auto J) NL_INL {
            constexpr int j = decltype(J)::value;
            float t = s;
            static_range<0, NC>([&](auto C) NL_INL {
                constexpr int c = decltype(C)::value;
                if ((ad >> c) & 1u) {
                } else if ((aa >> c) & 1u) {
                    NL_KEEP_BRANCH;
                    static_range<c * CH, c * CH + CH>([&](auto K) NL_INL { t = __fadd_rn(t, v[decltype(K)::value]); });
                } else {
                    NL_KEEP_BRANCH;
                    static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                        constexpr int k = decltype(K)::value;
                        t = __fadd_rn(t, NL_AND(v[k], NL_M(k)));
                    });
                }
            });
            s = quad_from<LPP, j>(t);
            forget_words<NW>(live);
        });
        const float ym = s / fm;
        // ---- variance and correlation sums (stats.go:254-257, 573-579) ----
        float vs = 0.0f, corr = 0.0f;
        static_range<0, LPP>([&](auto J) NL_INL {
            constexpr int j = decltype(J)::value;
            float tv = vs, tc = corr, fi = fi0;
            const float ym2 = opaque_f(ym);
            static_range<0, NC>([&](auto C) NL_INL {
                constexpr int c = decltype(C)::value;
                if ((ad >> c) & 1u) {
                } else if ((aa >> c) & 1u) {
                    NL_KEEP_BRANCH;
                    const float dxb = __fsub_rn(fi, xm);          // exact, see the one-lane kernel
                    static_range<0, CH>([&](auto U) NL_INL {
                        constexpr int u = decltype(U)::value;
                        const float dy = __fsub_rn(v[c * CH + u], ym2);
                        tv = __fadd_rn(tv, __fmul_rn(dy, dy));
                        tc = __fadd_rn(tc, __fmul_rn(__fadd_rn(dxb, (float)u), dy));
                    });
                    fi += (float)CH;
                } else {
                    NL_KEEP_BRANCH;
                    static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                        constexpr int k = decltype(K)::value;
                        const int lm = NL_M(k);
                        const float dy = __fsub_rn(v[k], ym2);
                        const float dd = __fmul_rn(dy, dy);
                        tv = __fadd_rn(tv, NL_AND(dd, lm));
                        const float dx = __fsub_rn(fi, xm);
                        const float t = __fmul_rn(dx, dy);
                        tc = __fadd_rn(tc, NL_AND(t, lm));
                        fi += NL_AND(1.0f, lm);
                    });
                }
            });
            vs = quad_from<LPP, j>(tv);
            corr = quad_from<LPP, j>(tc);
            forget_words<NW>(live);
        });
        const float ysd = sqrt_go(vs / fm);
        float den = __fmul_rn(xsd, ysd);
        den = __fmul_rn(den, __fadd_rn(fm, 1.0f));
        corr = corr / den;
        float slope = __fmul_rn(corr, ysd);
        slope = slope / xsd;
        float icpt = __fsub_rn(ym, __fmul_rn(slope, xm));
        // ---- mean absolute deviation from the fit (stack.go:879-886) ----
        float sg = 0.0f;
        float dmax[NW], dmin[NW];               // extremes of the residuals of the all-alive chunks, per 32 positions
        static_range<0, NW>([&](auto W) NL_INL { dmax[decltype(W)::value] = -__builtin_inff(); dmin[decltype(W)::value] = __builtin_inff(); });
        static_range<0, LPP>([&](auto J) NL_INL {
            constexpr int j = decltype(J)::value;
            float t = sg, fi = fi0;
            const float sl2 = opaque_f(slope);
            static_range<0, NC>([&](auto C) NL_INL {
                constexpr int c = decltype(C)::value;
                if ((ad >> c) & 1u) {
                } else if ((aa >> c) & 1u) {
                    NL_KEEP_BRANCH;
                    float dprev = 0.0f;
                    static_range<0, CH>([&](auto U) NL_INL {
                        constexpr int u = decltype(U)::value;
                        const float lin = __fadd_rn(__fmul_rn(__fadd_rn(fi, (float)u), sl2), icpt);
                        const float diff = __fsub_rn(v[c * CH + u], lin);
                        t = __fadd_rn(t, fabsf(diff));
                        if constexpr (j == 0) {            // the residuals do not depend on the chain
                            if constexpr ((u & 1) == 0) {
                                dprev = diff;
                            } else {
                                dmax[c / CPW] = max3_asm(dmax[c / CPW], dprev, diff);
                                dmin[c / CPW] = min3_asm(dmin[c / CPW], dprev, diff);
                            }
                        }
                    });
                    fi += (float)CH;
                } else {
                    NL_KEEP_BRANCH;
                    static_range<c * CH, c * CH + CH>([&](auto K) NL_INL {
                        constexpr int k = decltype(K)::value;
                        const int lm = NL_M(k);
                        const float lin = __fadd_rn(__fmul_rn(fi, sl2), icpt);
                        const float diff = __fsub_rn(v[k], lin);
                        t = __fadd_rn(t, NL_AND(fabsf(diff), lm));
                        fi += NL_AND(1.0f, lm);
                    });
                }
            });
            sg = quad_from<LPP, j>(t);
            forget_words<NW>(live);
        });
        sg = sg / fm;
        // ---- reject (stack.go:890-904): no chain, every lane does its own ranks ----
        float lb = __fmul_rn(p.sig_lo, sg), hb = __fmul_rn(p.sig_hi, sg);
        const bool bad = !(slope == slope) || !(icpt == icpt) || !(lb == lb) || !(hb == hb);
        if (bad) { slope = 0.0f; icpt = 0.0f; lb = __builtin_inff(); hb = __builtin_inff(); }
        unsigned holed = 0;                     // words in which a sample of an all-alive chunk is rejected in some lane
        static_range<0, NW>([&](auto W) NL_INL {
            constexpr int w = decltype(W)::value;
            holed |= (__any(active && !bad && (dmin[w] < -lb || dmax[w] > hb)) ? 1u : 0u) << w;
        });
        holed = (unsigned)__builtin_amdgcn_readfirstlane((int)holed);
        unsigned lo_n = 0, hi_n = 0;
        unsigned nlive[NW];
        static_range<0, NW>([&](auto W) NL_INL { nlive[decltype(W)::value] = live[decltype(W)::value]; });
        float fi = fi0;
        slope = opaque_f(slope);
        static_range<0, NC>([&](auto C) NL_INL {
            constexpr int c = decltype(C)::value;
            if ((ad >> c) & 1u) {
            } else if (((aa >> c) & 1u) && !((holed >> (c / CPW)) & 1u)) {
                NL_KEEP_BRANCH;
                fi += (float)CH;
            } else {
                NL_KEEP_BRANCH;
                unsigned lowb = 0, highb = 0;
                static_range<0, CH>([&](auto U) NL_INL {
                    constexpr int k = c * CH + decltype(U)::value;
                    const float lin = __fadd_rn(__fmul_rn(fi, slope), icpt);
                    const float t = __fsub_rn(lin, v[k]);
                    lowb = __builtin_amdgcn_alignbit(lowb, (unsigned)__float_as_int(__fsub_rn(lb, t)), 31);
                    highb = __builtin_amdgcn_alignbit(highb, (unsigned)__float_as_int(__fadd_rn(hb, t)), 31);
                    fi += NL_AND(1.0f, NL_M(k));
                });
                constexpr int sh = CH * (c % CPW);
                constexpr unsigned full = (1u << CH) - 1u;
                const unsigned alive = (live[c / CPW] >> sh) & full;
                const unsigned low = (__builtin_bitreverse32(lowb) >> (32 - CH)) & alive;
                const unsigned high = (__builtin_bitreverse32(highb) >> (32 - CH)) & alive & ~low;
                lo_n = opaque_u(lo_n + (unsigned)__popc(low));
                hi_n = opaque_u(hi_n + (unsigned)__popc(high));
                nlive[c / CPW] = opaque_u(nlive[c / CPW] & ~((low | high) << sh));
            }
        });
#undef NL_M
#undef NL_AND
        const int lo_all = quad_sum<LPP>((int)lo_n), hi_all = quad_sum<LPP>((int)hi_n);
        if (active) {
            p_lo += lo_all;
            p_hi += hi_all;
            const int left = lo_all + hi_all;
            res = ym;                                       // stack.go:911
            if (left == 0 || m < 3) active = false;
            m -= left;
            static_range<0, NW>([&](auto W) NL_INL { live[decltype(W)::value] = nlive[decltype(W)::value]; });
        }
    }

    const bool rep = on && role == 0;
    const bool more = active;                      // not done within this stage's quota (same in all lanes of the pixel)
    if (rep && !to_exact && !more) p.out[pix] = res;
    if (rep && !to_exact) { c_lo += p_lo; c_hi += p_hi; }
    const unsigned long long mm = __ballot(rep && more);
    if (__any(more)) {
        unsigned base = 0;
        if (lane == 0 && mm) base = atomicAdd(g.out_count, (unsigned)__popcll(mm));
        base = __shfl(base, 0, 64);
        int slot = (int)(base + (unsigned)__popcll(mm & ((1ull << lane) - 1ull)));     // valid in the pixel's lane 0
        slot = __float_as_int(quad_from<LPP, 0>(__int_as_float(slot)));
        if (more && (unsigned)slot < g.out_capacity) {
            if (role == 0) g.out_list[slot] = (unsigned)pix;
            g.out_state[(int64_t)slot * LPP + role] = make_uint4(live[0], live[1], live[2], live[3]);
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
        c_lo += __shfl_xor(c_lo, o, 64);
        c_hi += __shfl_xor(c_hi, o, 64);
    }
    if (lane == 0) { s_lo[threadIdx.x >> 6] = c_lo; s_hi[threadIdx.x >> 6] = c_hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t_lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
        const int t_hi = s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3];
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}

int linfit_ml_supported(int mode, int n_frames, int64_t npix)
{
    return (mode == NL_ST_LINEAR_FIT && n_frames > 128 && n_frames <= 512 && npix < ((int64_t)1 << 27)) ? 1 : 0;
}

// fit iterations per cascade stage (the last stage runs to the end); NL_LF_QUOTA="a,b,c" overrides.
// 8 / 6 / 8 measured best on 128 x 4096^2 in round 2 (19.7 ms vs 20.2 (6,6,8), 20.4 (10,8,8), 21.1 (5,5,8)) and again in round 6
// (flat within 1 % between 8,6,8 / 6,6,8 / 6,4,6), and best up to 40 frames (32 frames: 2.68 ms, 6,6,8: 3.50 -- a shallow
// continuation stage is a poor trade).  In between the first stage wants to stop earlier (round 6, profiles/r06_linfit_quota.txt):
// 48 frames 4.31 -> 3.98 ms with 7,6,8; 56 / 64 / 72 / 80 frames 5.46 / 6.07 / 8.32 / 9.13 -> 4.90 / 5.71 / 8.03 / 8.99 with 6,6,8;
// 96 frames 10.96 -> 10.84 with 7,6,8.
static const int *linfit_quota(int n_frames)
{
    static int env_quota[kLinfitStages] = {0, 0, 0, 0};
    static const bool from_env = [] {
        if (const char *e = getenv("NL_LF_QUOTA")) {
            int a = 0, b = 0, c = 0;
            if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) {
                env_quota[0] = a; env_quota[1] = b; env_quota[2] = c;
                return true;
            }
        }
        return false;
    }();
    if (from_env) return env_quota;
    static const int q868[kLinfitStages] = {8, 6, 8, 0}, q768[kLinfitStages] = {7, 6, 8, 0}, q668[kLinfitStages] = {6, 6, 8, 0};
    if (n_frames <= 40 || n_frames > 96) return q868;
    if (n_frames <= 48 || n_frames > 80) return q768;
    return q668;
}

template <int LPP>
static void launch_lf_ml(const StackArgs &args, const FastArgs &f, const LinfitCascade *c, hipStream_t stream,
                         hipEvent_t dominant_done)
{
    const unsigned per_wg = 256 / LPP;
    const unsigned blocks = (unsigned)((args.npix + per_wg - 1) / per_wg);
    LinfitStage g = {};
    if (!c) {
        hipLaunchKernelGGL((stack_linfit_ml_kernel<LPP, false>), dim3(blocks), dim3(256), 0, stream, args, f, g);
        if (dominant_done) (void)hipEventRecord(dominant_done, stream);
        return;
    }
    const int *quota = linfit_quota(args.n_frames);             // as the one-lane kernel
    for (int s = 0; s < kLinfitStages; s++) {
        g.max_iters = quota[s];
        g.in_list = s ? c->list[(s - 1) & 1] : nullptr;
        g.in_state = s ? c->state[(s - 1) & 1] : nullptr;
        g.in_count = s ? c->count + (s - 1) : nullptr;
        g.in_capacity = c->capacity;
        g.out_list = c->list[s & 1];
        g.out_state = c->state[s & 1];
        g.out_count = c->count + s;
        g.out_capacity = c->capacity;
        if (s == 0) {
            hipLaunchKernelGGL((stack_linfit_ml_kernel<LPP, false>), dim3(blocks), dim3(256), 0, stream, args, f, g);
            if (dominant_done) (void)hipEventRecord(dominant_done, stream);
        } else {
            const unsigned gblocks = blocks < 16384u ? blocks : 16384u;
            hipLaunchKernelGGL((stack_linfit_ml_kernel<LPP, true>), dim3(gblocks), dim3(256), 0, stream, args, f, g);
        }
    }
}

// cascade->state must hold LPP entries per listed pixel (2 up to 256 frames, else 4)
hipError_t launch_stack_linfit_ml(const StackArgs &args, const FastArgs &fargs, const LinfitCascade *cascade,
                                  hipStream_t stream, const char **name, hipEvent_t dominant_done)
{
    if (args.n_frames <= 2 * kMlNS) {
        *name = "stack_linfit_ml_kernel<2, false>";
        launch_lf_ml<2>(args, fargs, cascade, stream, dominant_done);
    } else {
        *name = "stack_linfit_ml_kernel<4, false>";
        launch_lf_ml<4>(args, fargs, cascade, stream, dominant_done);
    }
    return hipGetLastError();
}

int linfit_fast_supported(int mode, int n_frames, int64_t npix)
{
    return (mode == NL_ST_LINEAR_FIT && n_frames >= 1 && n_frames <= 128 && npix < kFastMaxPixels) ? 1 : 0;
}

template <int NS>
static void launch_lf(const StackArgs &args, const FastArgs &f, const LinfitCascade *c, unsigned blocks,
                      hipStream_t stream, hipEvent_t dominant_done)
{
    LinfitStage g = {};
    if (!c) {                                    // no cascade buffers: one stage, run to completion
        hipLaunchKernelGGL((stack_linfit_fast_kernel<NS, false>), dim3(blocks), dim3(256), 0, stream, args, f, g);
        if (dominant_done) (void)hipEventRecord(dominant_done, stream);
        return;
    }
    // Fit iterations per stage.  The number a pixel needs varies a lot (8 on average, 20-26
    // for the slowest lane of a wave): capping a stage and re-packing the unfinished pixels
    // into full waves halves the lane-iterations, at the price of re-sorting those pixels.
    const int *quota = linfit_quota(args.n_frames);
    for (int s = 0; s < kLinfitStages; s++) {
        g.max_iters = quota[s];
        g.in_list = s ? c->list[(s - 1) & 1] : nullptr;
        g.in_state = s ? c->state[(s - 1) & 1] : nullptr;
        g.in_count = s ? c->count + (s - 1) : nullptr;
        g.in_capacity = c->capacity;
        g.out_list = c->list[s & 1];
        g.out_state = c->state[s & 1];
        g.out_count = c->count + s;
        g.out_capacity = c->capacity;
        if (s == 0) {
            hipLaunchKernelGGL((stack_linfit_fast_kernel<NS, false>), dim3(blocks), dim3(256), 0, stream, args, f, g);
            if (dominant_done) (void)hipEventRecord(dominant_done, stream);
        } else {
            const unsigned gblocks = blocks < 8192u ? blocks : 8192u;
            hipLaunchKernelGGL((stack_linfit_fast_kernel<NS, true>), dim3(gblocks), dim3(256), 0, stream, args, f, g);
        }
    }
}

// one continuation stage of the one-lane kernel over stage.in_list (the bit-exact tail of other cascades)
void launch_linfit_exact_stage(const StackArgs &args, const FastArgs &fargs, const LinfitStage &stage, unsigned blocks,
                               hipStream_t stream)
{
    const int n = args.n_frames;
    if (n <= 8)        hipLaunchKernelGGL((stack_linfit_fast_kernel<8, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
    else if (n <= 16)  hipLaunchKernelGGL((stack_linfit_fast_kernel<16, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
    else if (n <= 32)  hipLaunchKernelGGL((stack_linfit_fast_kernel<32, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
    else if (n <= 48)  hipLaunchKernelGGL((stack_linfit_fast_kernel<48, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
    else if (n <= 64)  hipLaunchKernelGGL((stack_linfit_fast_kernel<64, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
    else if (n <= 96)  hipLaunchKernelGGL((stack_linfit_fast_kernel<96, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
    else               hipLaunchKernelGGL((stack_linfit_fast_kernel<128, true>), dim3(blocks), dim3(256), 0, stream, args, fargs, stage);
}

hipError_t launch_stack_linfit_fast(const StackArgs &args, const FastArgs &fargs, const LinfitCascade *cascade,
                                    hipStream_t stream, const char **name, hipEvent_t dominant_done)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    if (n <= 8)        { *name = "stack_linfit_fast_kernel<8, false>";   launch_lf<8>(args, fargs, cascade, blocks, stream, dominant_done); }
    else if (n <= 16)  { *name = "stack_linfit_fast_kernel<16, false>";  launch_lf<16>(args, fargs, cascade, blocks, stream, dominant_done); }
    else if (n <= 32)  { *name = "stack_linfit_fast_kernel<32, false>";  launch_lf<32>(args, fargs, cascade, blocks, stream, dominant_done); }
    else if (n <= 48)  { *name = "stack_linfit_fast_kernel<48, false>";  launch_lf<48>(args, fargs, cascade, blocks, stream, dominant_done); }
    else if (n <= 64)  { *name = "stack_linfit_fast_kernel<64, false>";  launch_lf<64>(args, fargs, cascade, blocks, stream, dominant_done); }
    else if (n <= 96)  { *name = "stack_linfit_fast_kernel<96, false>";  launch_lf<96>(args, fargs, cascade, blocks, stream, dominant_done); }
    else               { *name = "stack_linfit_fast_kernel<128, false>"; launch_lf<128>(args, fargs, cascade, blocks, stream, dominant_done); }
    return hipGetLastError();
}

}  // namespace nl
