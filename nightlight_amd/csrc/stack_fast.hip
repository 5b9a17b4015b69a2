// stack_fast.hip -- register-resident sigma-clip kernels for gfx950 (MI355X).
//
// One pixel per lane, the pixel's N samples in VGPRs (static indices only).
// A sorting network orders them once; after that the median is a lookup,
// clipping removes a prefix / suffix of the sorted column, and every later
// iteration of StackSigma (internal/ops/stack/stack.go:401-431) is a couple of
// passes over registers -- no LDS, no data-dependent addressing, no divergence
// inside a pass.  HBM traffic: each sample is read exactly once (4*(N+1) B per
// output pixel), 256 contiguous bytes per wave per frame.
//
// Exactness contract.  The reference computes mean / stddev as sequential fp32
// sums over a data-dependent permutation (qsort.go:94-126 leaves it, stats.go
// :246-261 sums over it), so its stddev cannot be reproduced bit for bit
// without replaying the permutation.  These kernels instead
//   * take the median exactly (order independent),
//   * compute mean~ / var~ in their own order and bracket the reference's
//     stddev in a rigorous interval [s_min, s_max] that holds for ANY summation
//     order (standard forward error bound),
//   * evaluate the reference's bound expressions median -/+ sigma*stddev at
//     both interval ends with the reference's own fp32 operations, and
//   * accept a clip decision only if it is the same at both ends.
// A pixel with any undecidable sample (a sample inside the few-ulp window, or
// an infinite sample) is appended to the "exact" list and re-done from scratch
// by the bit-exact kernel (stack_exact.hip).  Hence: clip counters are
// identical to the reference's, and the output mean differs only by summation
// order (<= ~1e-6 relative).
//
// Two instantiations per network size.  ZONAL: valid while every lane of the
// wave misses at most KP samples and has clipped fewer than KZ samples per side
// (8 and 8; 4 and 4 for networks below 48); then only the KZ lowest and KZ+KP highest sorted
// positions can ever be excluded and all other positions are summed without a
// mask.  A wave that leaves this regime (NaN borders, heavy clipping) puts its
// pixels on the "generic" list; the GENERIC instantiation re-does those from
// the list with every position masked by its rank.
#include <string>

#include "fast_common.hpp"

namespace nl {

// StackMedian (stack.go:274-303): the median is order independent, so the
// sorted register column gives it exactly (qsort.go:68-82: odd n -> middle,
// even n -> 0.5*(lower+upper)).  Bit-exact.
// WINDOW = true: grid covers the tile.  Only the ranks the median can occupy
//   while at most kMedianPad samples are missing need to be exact, the rest of
//   the network is pruned (ZonalNetwork: ~17 % fewer comparators) and the lookup
//   scans 10 registers instead of all; lanes with more missing samples go to
//   the hand-over list.
// WINDOW = false: full sort, any number of missing samples; over the whole tile
//   (small networks) or grid-stride over the hand-over list.
constexpr int kMedianPad = 16;

template <int NS, bool WINDOW>
__global__ __launch_bounds__(256) void stack_median_fast_kernel(StackArgs p, FastArgs q)
{
    constexpr int W0 = WINDOW ? (NS - kMedianPad) / 2 - 1 : 0, W1 = WINDOW ? NS / 2 + 1 : NS;
    using Sorter = std::conditional_t<WINDOW, ZonalSort<0, W0, W1, NS>, FullSort>;
    const bool listed = !WINDOW && q.in_list != nullptr;
    const int64_t limit = listed ? (int64_t)min(*q.in_count, q.in_capacity) : p.npix;
    const int64_t sweep = listed ? (int64_t)gridDim.x * blockDim.x : limit;
    const int lane = threadIdx.x & 63;
    // (the windowed instantiation is a single trip; saying so keeps the compiler from
    // overlapping two trips' columns in registers)
    for (int64_t wg_item = (int64_t)blockIdx.x * blockDim.x; wg_item < limit; wg_item += sweep) {
        int N = p.n_frames;
        asm volatile("" : "+s"(N));              // per trip: keeps the per-frame scalars out of the loop preheader
        const int64_t item = wg_item + threadIdx.x;
        const bool on = item < limit;
        int64_t pix = item;
        if (listed) pix = on ? (int64_t)q.in_list[item] : 0;
        const unsigned boff = (unsigned)(on ? pix : 0) * 4u;
        float v[NS];
        const int n = gather_sorted<NS, 16, Sorter, true>(p.frames, p.stride, N, boff, v);
        const int kk = n >> 1;
        bool hand_over = false;
        float upper, lower;
        if constexpr (WINDOW) {
            hand_over = on && n < NS - kMedianPad;          // kk-1 >= W0 and kk < W1 otherwise
            pick_pair<W0, W1>(v, kk, lower, upper);
        } else {
            pick_pair<0, NS>(v, kk, lower, upper);
        }
        float res = (n & 1) ? upper : 0.5f * (lower + upper);
        if (n == 0) res = p.ref_loc;
        if constexpr (WINDOW) {
            // Unconditional buffer store; lanes that must not write get an out-of-range offset,
            // which the hardware drops.  (A store under `if (on)` makes the compiler keep a
            // second copy of the column: 262 instead of 151 registers at NS = 128.)
            const __amdgpu_buffer_rsrc_t ors =
                __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(p.npix * 4), 0x00020000);     // npix < 2^29 (dispatch)
            const unsigned so = (on && !hand_over) ? (unsigned)pix * 4u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(res), ors, (int)so, 0, 0);
        } else {
            if (on) p.out[pix] = res;
        }
        if constexpr (WINDOW) {
            const unsigned long long gm = __ballot(hand_over);
            if (gm) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(q.gen_count, (unsigned)__popcll(gm));
                base = __shfl(base, 0, 64);
                const unsigned slot = base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull));
                if (hand_over && slot < q.gen_capacity) q.gen_list[slot] = (unsigned)pix;
            }
            break;
        }
    }
}

// StackMADSigma (stack.go:536-605), register resident.  The clip bounds come from two
// medians (of the samples, and of their absolute deviations from it), both order
// independent, so the bounds, every clip decision and both counters are exact without
// any guard; only the final mean is an order dependent fp32 sum (the reference adds the
// survivors in the order two quickselects and the clip swaps left them), which this
// kernel forms in frame order: <= summation-order rounding, like the sigma kernel.
//   1. gather + sort            -> median (lookup)
//   2. column := |x - median|   -> sort again -> MAD (lookup); the samples are gone
//   3. second read of the pixel's frames (L2 / MALL resident), in frame order:
//      count x < lo, x > hi, sum the rest
// A pixel whose median is not finite (half of its samples infinite) has NaN deviations
// (Inf - Inf); what the reference's quickselect makes of those depends on where they sit
// (for some inputs it runs off the array and panics).  Such a pixel goes to the exact
// kernel, which follows the reference step by step where that is defined.
template <int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS > 80 ? 3 : 1, 8)))   // (96: 190 VGPRs otherwise, since the 128-position merge)
void stack_mad_fast_kernel(StackArgs p, FastArgs q)
{
    // q.in_list: the pixels stack_mad_bitonic_kernel handed over (the list's length is only known on
    // the device: fixed grid, grid-stride loop); otherwise the grid covers the tile
    const int64_t limit = q.in_list ? (int64_t)min(*q.in_count, q.in_capacity) : p.npix;
    const int lane = threadIdx.x & 63;
    int c_lo_sum = 0, c_hi_sum = 0;
  for (int64_t item0 = (int64_t)blockIdx.x * blockDim.x; item0 < limit; item0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t item = item0 + threadIdx.x;
    const bool on = item < limit;
    int64_t pix = item;
    if (q.in_list) pix = on ? (int64_t)q.in_list[item] : 0;
    const unsigned boff = (unsigned)(on ? pix : 0) * 4u;
    int N = p.n_frames;
    asm volatile("" : "+s"(N));
    float v[NS];
    const int n = gather_sorted<NS, 16, FullSortT<false>>(p.frames, p.stride, N, boff, v);
    const int kk = n >> 1;                                   // qsort.go:70: k = (n>>1)+1, 1-based
    float upper, lower;
    pick_pair<0, NS>(v, kk, lower, upper);
    const float median = (n & 1) ? upper : 0.5f * (lower + upper);
    const bool degenerate = n > 0 && !(__builtin_fabsf(median) < __builtin_inff());
    const float msafe = degenerate ? 0.0f : median;
    float dupper, dlower;
    if constexpr (NS > 64) {
        // the deviations of a sorted column fall to the median and rise again (pads: +Inf on top): a
        // bitonic sequence, which the half-cleaner cascade of a bitonic merge sorts -- 626 fused
        // operations on 128 positions (FusedBitonic, sort_tables.inc) instead of a second sorting
        // network (1 500 .. 2 184)
        float d[128];
        static_chunks<0, 128, 16>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            if constexpr (k < NS) d[k] = __builtin_fabsf(v[k] - msafe);          // stack.go:566-571 (pads stay +Inf)
            else                  d[k] = __builtin_inff();
        });
        run_network<FusedBitonic<128, 0>, 128>(d);
        pick_pair<0, NS>(d, kk, dlower, dupper);
    } else {
        static_chunks<0, NS, 16>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            v[k] = __builtin_fabsf(v[k] - msafe);            // stack.go:566-571 (pads stay +Inf)
        });
        sort_network<NS, false>(v);
        pick_pair<0, NS>(v, kk, dlower, dupper);
    }
    const float mad = (n & 1) ? dupper : 0.5f * (dlower + dupper);
    const float sd = mad * 1.4826f;                          // stack.go:574
    const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
    const float lo = median - t_lo, hi = median + t_hi;

    // second read, frame order: all loads first, into the column's registers (free again)
    // (frame count and pitch re-read through opaque registers: otherwise the scalar offsets and
    // descriptors of the first gather are kept alive across both sorts for re-use here)
    int N2 = p.n_frames;
    asm volatile("" : "+s"(N2));
    int64_t frame_bytes = p.stride * (int64_t)sizeof(float);
    asm volatile("" : "+s"(frame_bytes));
    const int last = N2 - 1;
    static_chunks<0, NS / 4, 4>([&](auto C) NL_INL {
        constexpr int c0 = 4 * decltype(C)::value;
        const int f0 = min(c0, last);
        const char *gb = reinterpret_cast<const char *>(p.frames) + (int64_t)f0 * frame_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gb), 0, -1, 0x00020000);
        static_range<0, 4>([&](auto U) NL_INL {
            constexpr int k = c0 + decltype(U)::value;
            const int soff = (min(k, last) - f0) * (int)frame_bytes;
            v[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)boff, soff, 0));
        });
    });
    // counts as floats (exact far beyond 128): integer sums would be re-associated into a tree,
    // which keeps one lane mask per sample alive
    float f_lo = 0.0f, f_hi = 0.0f, f_kept = 0.0f, sum = 0.0f;
    static_chunks<0, NS, 4>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        const float x = v[k];
        const bool present = (k < N2) && (x == x);           // NaN = no data, positions past N unused
        const bool below = present && x < lo;                // stack.go:583-592: low first
        const bool above = present && !below && x > hi;
        const bool keep = present && !below && !above;
        f_lo += below ? 1.0f : 0.0f;
        f_hi += above ? 1.0f : 0.0f;
        f_kept += keep ? 1.0f : 0.0f;
        sum += keep ? x : 0.0f;
        // (pinned here: otherwise the four chains are sunk below all compares and every
        // sample's lane masks are parked in SGPRs until then)
        asm volatile("" : "+v"(f_lo), "+v"(f_hi), "+v"(f_kept), "+v"(sum));
    });
    int c_lo = (int)f_lo, c_hi = (int)f_hi;
    float res = sum / f_kept;                                // no survivor: 0/0 = NaN, as the reference
    if (n == 0) res = p.ref_loc;
    const bool to_exact = on && degenerate;
    if (on && !to_exact) p.out[pix] = res;
    if (!on || to_exact || n == 0) { c_lo = 0; c_hi = 0; }
    c_lo_sum += c_lo;
    c_hi_sum += c_hi;
    const unsigned long long em = __ballot(to_exact);
    if (em) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
        if (to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
    }
  }
    int c_lo = c_lo_sum, c_hi = c_hi_sum;
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

// 128 frames: no second sort and no second read.  The deviations |x - median| of a SORTED column
// fall to the median and rise again -- a bitonic sequence (rounding is monotone; pads stay +Inf at
// the top) -- so the half-cleaner cascade of a bitonic merge would sort them, and only the ranks
// the MAD can occupy are wanted (56..71 for n >= 114 samples): L_i = min(d_i, d_i+64) are the 64
// smallest deviations, U_i = max(d_i, d_i+64) the 64 largest; the maxima of the L_i over
// i = j mod 8 are, once sorted, ranks 56..63, the minima of the U_i ranks 64..71.  About 370
// instructions on the fly instead of a 2184-instruction sort of a second column, and the samples
// stay in their registers for the clip and the mean (summed in sorted order: the reference sums
// in the order its quickselects left -- summation-order rounding either way).  Pixels with fewer
// than 114 samples go to q.gen_list, which stack_mad_fast_kernel<128> finishes.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void stack_mad_bitonic_kernel(StackArgs p, FastArgs q)
{
    constexpr int NS = 128;
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = pix < p.npix;
    const int lane = threadIdx.x & 63;
    const unsigned boff = (unsigned)(on ? pix : 0) * 4u;
    int N = p.n_frames;
    asm volatile("" : "+s"(N));
    float v[NS];
    const int n = gather_sorted<NS, 16, FullSortT<false>>(p.frames, p.stride, N, boff, v);
    const bool narrow = on && n > 0 && n < 114;
    const int kk = min(max(n >> 1, 57), 64);                 // qsort.go:70: k = (n>>1)+1, 1-based (clamped for lanes that leave)
    float upper, lower;
    pick_pair<56, 65>(v, kk, lower, upper);
    const float median = (n & 1) ? upper : 0.5f * (lower + upper);
    const bool degenerate = n > 0 && !(__builtin_fabsf(median) < __builtin_inff());
    const float msafe = degenerate ? 0.0f : median;
    float win[16];
    static_range<0, 8>([&](auto J) NL_INL {
        constexpr int j = decltype(J)::value;
        float w = -__builtin_inff(), u = __builtin_inff();
        static_range<0, 8>([&](auto T) NL_INL {
            constexpr int i = j + 8 * decltype(T)::value;
            const float dl = __builtin_fabsf(v[i] - msafe), dr = __builtin_fabsf(v[i + 64] - msafe);   // stack.go:566-571
            w = fmaxf(w, fminf(dl, dr));
            u = fminf(u, fmaxf(dl, dr));
        });
        win[j] = w;
        win[8 + j] = u;
    });
    // two bitonic runs of 8: distances 4, 2, 1
    static_range<0, 3>([&](auto S) NL_INL {
        constexpr int dist = 4 >> decltype(S)::value;
        static_range<0, 16>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            if constexpr ((i & dist) == 0) {
                const float a = win[i], b = win[i + dist];
                win[i] = fminf(a, b);
                win[i + dist] = fmaxf(a, b);
            }
        });
    });
    float dupper, dlower;
    pick_pair<0, 16>(win, kk - 56, dlower, dupper);          // ranks kk-1, kk of the deviations
    const float mad = (n & 1) ? dupper : 0.5f * (dlower + dupper);
    const float sd = mad * 1.4826f;                          // stack.go:574
    const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
    const float lo = median - t_lo, hi = median + t_hi;
    float f_lo = 0.0f, f_hi = 0.0f, f_kept = 0.0f, sum = 0.0f;
    int nn = n;
    static_chunks<0, NS, 4>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        const float x = v[k];
        if constexpr ((k & 7) == 0) nn = opaque(nn);
        const bool present = k < nn;                         // pads (+Inf) sort last
        const bool below = present && x < lo;                // stack.go:583-592: low first
        const bool above = present && !below && x > hi;
        const bool keep = present && !below && !above;
        f_lo += below ? 1.0f : 0.0f;
        f_hi += above ? 1.0f : 0.0f;
        f_kept += keep ? 1.0f : 0.0f;
        sum += keep ? x : 0.0f;
        asm volatile("" : "+v"(f_lo), "+v"(f_hi), "+v"(f_kept), "+v"(sum));
    });
    int c_lo = (int)f_lo, c_hi = (int)f_hi;
    float res = sum / f_kept;                                // no survivor: 0/0 = NaN, as the reference
    if (n == 0) res = p.ref_loc;
    const bool to_exact = on && degenerate && !narrow;
    if (on && !to_exact && !narrow) p.out[pix] = res;
    if (!on || to_exact || narrow || n == 0) { c_lo = 0; c_hi = 0; }
    const unsigned long long em = __ballot(to_exact), gm = __ballot(narrow);
    if (em) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(q.fb_count, (unsigned)__popcll(em));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
        if (to_exact && slot < q.fb_capacity) q.fb_list[slot] = (unsigned)pix;
    }
    if (gm) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(q.gen_count, (unsigned)__popcll(gm));
        base = __shfl(base, 0, 64);
        const unsigned slot = base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull));
        if (narrow && slot < q.gen_capacity) q.gen_list[slot] = (unsigned)pix;
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

// ZONAL = true : grid covers the tile, lane = pixel blockIdx*256+thread;
// ZONAL = false: grid-stride over q.in_list (pixels handed over by the zonal
//                kernel), any number of missing / clipped samples.
// TIGHT (zonal only): the stack has exactly NS frames, so the only missing samples are a
//                pixel's own NaNs -- the high zone reserves no positions for them (a third
//                fewer zone positions to mask, count and re-sum every clipping round); a lane
//                whose NaNs leave no survivor in the high zone goes to the generic pass as before.
#ifdef NL_ROUND_STATS
// developer statistics (build with make EXTRA=-DNL_ROUND_STATS): [0] winsor rounds executed by waves,
// [1] rounds lanes needed, [2] clip passes executed by waves, [3] clip passes lanes needed, [4] waves
__device__ unsigned long long nl_dbg_rounds[8];
extern "C" int nl_debug_round_stats(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nl_dbg_rounds), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nl_dbg_rounds), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define NL_STAT(i, x) atomicAdd(&nl_dbg_rounds[i], (unsigned long long)(x))
#else
#define NL_STAT(i, x) ((void)0)
#endif

template <int NS, bool ZONAL, bool WINSOR, bool TIGHT>
__global__ __launch_bounds__(256) void stack_sigma_fast_kernel(StackArgs p, FastArgs q)
{
    static_assert(!TIGHT || ZONAL, "TIGHT is a variant of the zonal kernels");
    if constexpr (ZONAL) fused_prologue_dominant(p);
    if constexpr (!ZONAL) { if (q.in_list) { fused_collect_slots(p); snapshot_fb_list(q); } }
    // zone widths: 8 clipped + 8 missing samples per lane for the larger
    // networks, 4 + 4 for the small ones
    constexpr int KZ = NS >= 48 ? kZone : 4, KP = TIGHT ? 0 : (NS >= 48 ? kPadMax : 4);
    static_assert(!ZONAL || NS >= 24, "zonal passes need room between the zones");
    constexpr int ZL = KZ;                                    // low zone  = positions [0, ZL)
    constexpr int ZH = ZONAL ? NS - KZ - KP : NS;             // high zone = positions [ZH, NS)

    int c_lo_total = 0, c_hi_total = 0;
    // ZONAL, or GENERIC without a list: the grid covers the tile, one pixel per
    // lane, a single trip.  GENERIC with a list: grid-stride over the hand-over
    // list, whose length is only known on the device.
    const bool listed = !ZONAL && q.in_list != nullptr;
    const int64_t limit = listed ? (int64_t)min(*q.in_count, q.in_capacity) : p.npix;
    const int64_t sweep = listed ? (int64_t)gridDim.x * blockDim.x : limit;
    const int lane = threadIdx.x & 63;

    for (int64_t wg_item = (int64_t)blockIdx.x * blockDim.x; wg_item < limit; wg_item += sweep) {
        // the frame count is re-read through an opaque register every trip:
        // otherwise the compiler hoists the 128 per-frame scalar selects that
        // depend on it out of the loop and spills them
        int N = p.n_frames;
        asm volatile("" : "+s"(N));
        const int64_t item = wg_item + threadIdx.x;
        const bool on = item < limit;
        int64_t pix = item;
        if (listed) pix = on ? (int64_t)q.in_list[item] : 0;
        const unsigned boff = (unsigned)(on ? pix : 0) * 4u;     // byte offset inside a frame

        // A genuine +-Inf sample stays among the n valid ones, makes the variance
        // non-finite and thereby sends the pixel to the exact kernel (`bail`).
        // zonal sigma: only the clip zones and the median window need exact ranks (the
        // winsorized variant also reads single positions in between: full sort)
        constexpr int MW0 = ZONAL ? ZH / 2 - 1 : 0, MW1 = ZONAL ? ZL + NS / 2 + 1 : NS;
        using Sorter = std::conditional_t<ZONAL && !WINSOR, ZonalSort<ZL, MW0, MW1, ZH>, FullSort>;
        float v[NS];
        const int n = gather_sorted<NS, 16, Sorter, true>(p.frames, p.stride, N, boff, v);
        bool to_exact = false;

        float res = p.ref_loc;
        int c_lo = 0, c_hi = 0;
        int a = 0, b = n;                       // surviving samples = sorted positions [a, b)
        bool active = on && n > 0;
        bool to_generic = false;
        if constexpr (ZONAL) {
            // zonal passes need b > ZH (and a < ZL): lanes with more missing
            // samples are handed to the generic pass, the others carry on
            to_generic = active && !(n > ZH);
            active = active && !to_generic;
        }

        // Shift c = first-pass median (any value near the bulk works).  With
        // D = sum(x-c) and Q = sum((x-c)^2) over the survivors,
        //     mean = c + D/cnt,   var = Q/cnt - (mean-c)^2        (exact identities).
        // Positions [ZL,ZH) are never clipped in the zonal passes, so their
        // share of D and Q is computed once; an iteration only re-sums the zones.
        constexpr int W0 = ZONAL ? ZH / 2 - 1 : 0, W1 = ZONAL ? ZL + NS / 2 + 1 : NS;
        const float c = pick<W0, W1>(v, a + ((b - a) >> 1));
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
#ifndef NL_WINSOR_WL
#define NL_WINSOR_WL(ns) ((ns) >= 112 ? 20 : ((ns) >= 80 ? 16 : ((ns) / 4 + 3) / 4 * 4))
#endif
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

        // max|x| over the survivors (only enters the reference-mean error term):
        // first pass from the two ends of the sorted column, afterwards from the
        // bounds every survivor passed
        float amax = fmaxf(fabsf(v[0]), fabsf(pick<ZONAL ? ZH : 0, NS>(v, n - 1)));

        if (ZONAL && lane == 0) NL_STAT(4, 1);
        while (__any(active)) {
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
            const float delta = dsum / fcnt;             // mean~ - c
            const float m = c + delta;
            const float aa = qsum / fcnt;                // E[(x-c)^2]~
            const float bb = delta * delta;
            const float var = fmaxf(aa - bb, 0.0f);

            // ---- bracket the reference's stddev ----
            // ours: aa carries <= NS/4+8 roundings per term; bb = delta^2 with delta off by
            // <= (NS/4+7) u mean|e|, and 2|delta| mean|e| <= aa + bb: together <= (NS/2+17) u (aa+bb)
            const float err_o = ((float)(NS / 2 + 24)) * kU * (aa + bb);
            // reference: relative gamma_(n+3) on its variance, its mean off by <= e_m
            const float eps_r = 1.02f * (fcnt + 8.0f) * kU;
            const float e_m = 1.02f * (fcnt + 2.0f) * kU * amax;
            const float v_up = var + err_o;
            const float v_dn = fmaxf(var - err_o, 0.0f);
            const float v_hi = v_up + v_up * eps_r + e_m * e_m;
            const float v_lo = fmaxf(v_dn - v_dn * eps_r, 0.0f);
            float s_max = __fsqrt_rn(v_hi) * (1.0f + 4.0f * kU);
            float s_min = __fsqrt_rn(v_lo) * (1.0f - 4.0f * kU);
            bool bail = !(v_hi < 3.0e38f);          // overflow / NaN (e.g. an Inf sample): exact kernel

            // ---- exact median (qsort.go:68-82): sorted column, position lookup ----
            // zonal: a in [0,ZL), b in (ZH,NS]  =>  kk in [ZH/2, ZL-1+NS/2]
            const int kk = a + (cnt >> 1);
            float upper, lower;
            pick_pair<W0, W1>(v, kk, lower, upper);
            const float median = (cnt & 1) ? upper : 0.5f * (lower + upper);

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
                wi.start(s_min, s_max);
                const float inv_cnt = 1.0f / fcnt;
                bool inner = active && !bail;
                while (__any(inner)) {
                    if (ZONAL) { if (lane == 0) NL_STAT(0, 1); if (inner) NL_STAT(1, 1); }
                    wi.next_clamp(median, xmin, xmax);
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
                    werr = ((float)(NS / 2 + 34)) * kU * (wa + wb);
                    wmean_c = wd;                                  // mean of the copy, minus c
                    wrms = wa;                                     // E[(copy - c)^2]
                    };
                    float var_t, err_t, wd_t, wa_t;
                    clamped_variance(wi.Lp, wi.Hm, var_t, err_t, wd_t, wa_t);
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
                        const float n_lo = (float)min(CS * t_lo + CS - 1, cnt), n_hi = (float)min(CS * t_hi + CS - 1, cnt);
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
                }
                s_min = wi.hull_lo;
                s_max = wi.hull_hi;
            }

            // ---- the reference's bound expressions at both ends of the interval ----
            // (stack.go:408-409; fp32 multiply then add, never fused)
            const float tl0 = __fmul_rn(p.sig_lo, s_min), tl1 = __fmul_rn(p.sig_lo, s_max);
            const float th0 = __fmul_rn(p.sig_hi, s_min), th1 = __fmul_rn(p.sig_hi, s_max);
            const float la = __fsub_rn(median, tl0), lb = __fsub_rn(median, tl1);
            const float ha = __fadd_rn(median, th0), hb = __fadd_rn(median, th1);
            const float lo_min = fminf(la, lb), lo_max = fmaxf(la, lb);
            const float hi_min = fminf(ha, hb), hi_max = fmaxf(ha, hb);

            // ---- count certain clips (c1,d1) and possible clips (c2,d2) ----
            // The column is sorted, so the samples below a threshold are a prefix and
            // those above it a suffix (pads are +Inf): count over the whole zone
            // without rank masks and subtract what is already excluded.
            int c1 = 0, c2 = 0, d1 = 0, d2 = 0;
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
                // the zones must still hold a survivor on each side, otherwise the
                // next sorted position (outside the zone) might be clipped as well:
                // such a lane restarts in the generic pass
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
                c1 = min(max(c1 - a, 0), cnt); c2 = min(max(c2 - a, 0), cnt);
                d1 = min(max(d1 - (NS - b), 0), cnt); d2 = min(max(d2 - (NS - b), 0), cnt);
            }
            if (active) {
                // a sample inside the window, or (negative sigma) inverted bounds where the
                // reference's "low first" order matters: let the exact kernel decide
                bail |= (c1 != c2) || (d1 != d2) || (lo_max > hi_min && (c1 + d1) > 0);
                if (bail) {
                    to_exact = true;
                    active = false;
                } else {
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
        }

        if (on && !to_generic && !to_exact) {
            p.out[pix] = res;
            c_lo_total += c_lo;
            c_hi_total += c_hi;
        }
        // hand-over lists: one atomic per wave reserves a contiguous run, lanes
        // fill it in lane order, so the consumer's loads stay coalesced
        if constexpr (ZONAL) {
            const unsigned long long gm = __ballot(on && to_generic);
            if (gm) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(q.gen_count, (unsigned)__popcll(gm));
                base = __shfl(base, 0, 64);
                const unsigned slot = base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull));
                if (on && to_generic && slot < q.gen_capacity) q.gen_list[slot] = (unsigned)pix;
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
        const int t_lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
        const int t_hi = s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3];
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if constexpr (!ZONAL) slot = clip_slot(p);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}

int fast_supported(int mode, bool weighted, int n_frames, int64_t npix)
{
    if (n_frames < 2 || n_frames > 128 || npix >= kFastMaxPixels) return 0;
    if (mode == NL_ST_MEDIAN) return 1;
    return ((mode == NL_ST_SIGMA || mode == NL_ST_WINSOR_SIGMA) && !weighted) ? 1 : 0;
}

template <int NS>
static void launch_median(const StackArgs &args, const FastArgs &fargs, unsigned blocks, hipStream_t stream,
                          const char **name, hipEvent_t dominant_done)
{
    static const std::string names[2] = {"stack_median_fast_kernel<" + std::to_string(NS) + ", false>",
                                         "stack_median_fast_kernel<" + std::to_string(NS) + ", true>"};
    FastArgs f = fargs;
    f.in_list = nullptr;
    f.in_count = nullptr;
    f.in_capacity = 0;
    const bool window = NS >= 32 && args.n_frames > NS - kMedianPad && fargs.gen_list != nullptr &&
                        args.npix < ((int64_t)1 << 29);
    if constexpr (NS >= 32) {
        if (window) {
            *name = names[1].c_str();
            hipLaunchKernelGGL((stack_median_fast_kernel<NS, true>), dim3(blocks), dim3(256), 0, stream, args, f);
            if (dominant_done) (void)hipEventRecord(dominant_done, stream);
            f.in_list = fargs.gen_list;
            f.in_count = fargs.gen_count;
            f.in_capacity = fargs.gen_capacity;
            const unsigned gblocks = blocks < kGenericGrid ? blocks : kGenericGrid;
            hipLaunchKernelGGL((stack_median_fast_kernel<NS, false>), dim3(gblocks), dim3(256), 0, stream, args, f);
            return;
        }
    }
    *name = names[0].c_str();
    hipLaunchKernelGGL((stack_median_fast_kernel<NS, false>), dim3(blocks), dim3(256), 0, stream, args, f);
    if (dominant_done) (void)hipEventRecord(dominant_done, stream);
}

hipError_t launch_stack_median_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                    const char **name, hipEvent_t dominant_done)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    if (n <= 8)        launch_median<8>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 16)  launch_median<16>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 32)  launch_median<32>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 48)  launch_median<48>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 64)  launch_median<64>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 80)  launch_median<80>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 96)  launch_median<96>(args, fargs, blocks, stream, name, dominant_done);
    else if (n <= 112) launch_median<112>(args, fargs, blocks, stream, name, dominant_done);
    else               launch_median<128>(args, fargs, blocks, stream, name, dominant_done);
    return hipGetLastError();
}


template <int NS>
static void launch_mad(const StackArgs &args, const FastArgs &f, unsigned blocks, hipStream_t stream, const char **name)
{
    static const std::string nm = "stack_mad_fast_kernel<" + std::to_string(NS) + ">";
    *name = nm.c_str();
    hipLaunchKernelGGL(stack_mad_fast_kernel<NS>, dim3(blocks), dim3(256), 0, stream, args, f);
}

int mad_fast_supported(int mode, bool weighted, int n_frames, int64_t npix)
{
    return (mode == NL_ST_MAD_SIGMA && !weighted && n_frames >= 1 && n_frames <= 128 && npix < kFastMaxPixels) ? 1 : 0;
}

hipError_t launch_stack_mad_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    if (n <= 8)        launch_mad<8>(args, fargs, blocks, stream, name);
    else if (n <= 16)  launch_mad<16>(args, fargs, blocks, stream, name);
    else if (n <= 32)  launch_mad<32>(args, fargs, blocks, stream, name);
    else if (n <= 48)  launch_mad<48>(args, fargs, blocks, stream, name);
    else if (n <= 64)  launch_mad<64>(args, fargs, blocks, stream, name);
    else if (n <= 80)  launch_mad<80>(args, fargs, blocks, stream, name);
    else if (n <= 96)  launch_mad<96>(args, fargs, blocks, stream, name);
    else if (n <= 112) launch_mad<112>(args, fargs, blocks, stream, name);
    else if (n < 114 || !fargs.gen_list) launch_mad<128>(args, fargs, blocks, stream, name);
    else {
        *name = "stack_mad_bitonic_kernel";
        FastArgs f = fargs;
        f.in_list = nullptr;
        f.in_count = nullptr;
        f.in_capacity = 0;
        hipLaunchKernelGGL(stack_mad_bitonic_kernel, dim3(blocks), dim3(256), 0, stream, args, f);
        // the pixels with fewer than 114 samples (aligned frames' borders): two sorts, second read
        f.in_list = fargs.gen_list;
        f.in_count = fargs.gen_count;
        f.in_capacity = fargs.gen_capacity;
        const unsigned gblocks = blocks < kGenericGrid ? blocks : kGenericGrid;
        hipLaunchKernelGGL(stack_mad_fast_kernel<128>, dim3(gblocks), dim3(256), 0, stream, args, f);
    }
    return hipGetLastError();
}

// smallest network size with a zonal instantiation
constexpr int kZonalMinSize = 24;

// kernel names as rocprofv3 prints them (template arguments: NS, ZONAL, WINSOR, TIGHT)
template <int NS, bool ZONAL, bool WINSOR, bool TIGHT>
static const char *sigma_kernel_name()
{
    static const std::string name = std::string("stack_sigma_fast_kernel<") + std::to_string(NS) + ", " +
                                    (ZONAL ? "true" : "false") + ", " + (WINSOR ? "true" : "false") + ", " +
                                    (TIGHT ? "true" : "false") + ">";
    return name.c_str();
}

// The nested launchers (stack_fast_mlg.hip, stack_fast_mlz.hip) end in hipGetLastError(), which CLEARS
// the pending error: their result is kept here (first error wins) so that a failed launch of the
// dominant kernel or of the generic pass reaches nl_stack_run instead of being erased.
static inline void keep_first(hipError_t &acc, hipError_t e) { if (acc == hipSuccess) acc = e; }

template <int NS, bool WINSOR>
static hipError_t launch_pair(const StackArgs &args, const FastArgs &fargs, unsigned tile_blocks,
                              hipStream_t stream, const char **name, hipEvent_t dominant_done,
                              AfterDominant after, void *user)
{
    hipError_t err = hipSuccess;
    FastArgs f = fargs;
    f.in_list = nullptr;
    f.in_count = nullptr;
    f.in_capacity = 0;
    if constexpr (NS >= kZonalMinSize) {
        if (args.n_frames == NS) {
            *name = sigma_kernel_name<NS, true, WINSOR, true>();
            hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, true, WINSOR, true>), dim3(tile_blocks), dim3(256), 0,
                               stream, args, f);
        } else {
            *name = sigma_kernel_name<NS, true, WINSOR, false>();
            hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, true, WINSOR, false>), dim3(tile_blocks), dim3(256), 0,
                               stream, args, f);
        }
        keep_first(err, hipGetLastError());
        if (dominant_done) keep_first(err, hipEventRecord(dominant_done, stream));
        if (after) after(user);
        // generic pass over the pixels the zonal waves handed over (its length
        // is only known on the device: fixed grid, grid-stride loop)
        f.in_list = fargs.gen_list;
        f.in_count = fargs.gen_count;
        f.in_capacity = fargs.gen_capacity;
        const unsigned gblocks = generic_grid(fargs.gen_hint, 256, tile_blocks < kGenericGrid ? tile_blocks : kGenericGrid);
        if constexpr (NS > 64) {
            // whole columns + prefix sums in LDS (stack_fast_mlg.hip): a clipping or winsorization round
            // is a few LDS reads instead of a pass over 128 masked registers -- this pass is pure
            // latency (a few hundred waves at most), and it sits on every pass's critical path
            const unsigned lblocks = generic_grid(fargs.gen_hint, 64, 4 * tile_blocks < 4 * kGenericGrid ? 4 * tile_blocks : 4 * kGenericGrid);
            keep_first(err, launch_stack_sigma_mlg(args, f, lblocks, stream, WINSOR));
        } else {
            hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, false, WINSOR, false>), dim3(gblocks), dim3(256), 0,
                               stream, args, f);
            keep_first(err, hipGetLastError());
        }
    } else {
        // small stacks: generic passes are cheap, run them over the whole tile
        *name = sigma_kernel_name<NS, false, WINSOR, false>();
        hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, false, WINSOR, false>), dim3(tile_blocks), dim3(256), 0,
                           stream, args, f);
        keep_first(err, hipGetLastError());
        if (dominant_done) keep_first(err, hipEventRecord(dominant_done, stream));
        if (after) after(user);
    }
    return err;
}

template <bool WINSOR>
static hipError_t launch_sized(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name,
                         hipEvent_t dominant_done, AfterDominant after, void *user)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    // network sizes: the frame count rounded up to the next instantiated size;
    // unused positions count as missing samples
    if (n <= 8)        return launch_pair<8, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 16)  return launch_pair<16, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 24)  return launch_pair<24, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 32)  return launch_pair<32, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 48)  return launch_pair<48, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 64)  return launch_pair<64, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 80)  return launch_pair<80, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 96)  return launch_pair<96, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else if (n <= 112) return launch_pair<112, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
    else               return launch_pair<128, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user);
}

hipError_t launch_stack_sigma_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                   const char **name, hipEvent_t dominant_done,
                                   bool winsor, AfterDominant after, void *user)
{
    hipError_t err = winsor ? launch_sized<true>(args, fargs, stream, name, dominant_done, after, user)
                            : launch_sized<false>(args, fargs, stream, name, dominant_done, after, user);
    keep_first(err, hipGetLastError());
    return err;
}

}  // namespace nl
