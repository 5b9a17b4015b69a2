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
#ifdef NL_PLAIN_STORES
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(res), ors, (int)so, 0, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(res), ors, (int)so, 0, 2);     // aux 2 = nt, see NL_STORE_RESULT
#endif
        } else {
            if (on) NL_STORE_RESULT(&p.out[pix], res);
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
    if (on && !to_exact) NL_STORE_RESULT(&p.out[pix], res);
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
    if (on && !to_exact && !narrow) NL_STORE_RESULT(&p.out[pix], res);
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

}  // namespace nl
#include "stack_fast_sigma_impl.hpp"
namespace nl {

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
constexpr int kZonalMinSize = 16;

// kernel names as rocprofv3 prints them (template arguments: NS, ZONAL, WINSOR, TIGHT, RECORD)
template <int NS, bool ZONAL, bool WINSOR, bool TIGHT>
static const char *sigma_kernel_name()
{
    static const std::string name = std::string("stack_sigma_fast_kernel<") + std::to_string(NS) + ", " +
                                    (ZONAL ? "true" : "false") + ", " + (WINSOR ? "true" : "false") + ", " +
                                    (TIGHT ? "true" : "false") + ", false, false>";
    return name.c_str();
}

// The nested launchers (stack_fast_mlg.hip, stack_fast_mlz.hip) end in hipGetLastError(), which CLEARS
// the pending error: their result is kept here (first error wins) so that a failed launch of the
// dominant kernel or of the generic pass reaches nl_stack_run instead of being erased.
static inline void keep_first(hipError_t &acc, hipError_t e) { if (acc == hipSuccess) acc = e; }

template <int NS, bool WINSOR>
static hipError_t launch_pair(const StackArgs &args, const FastArgs &fargs, unsigned tile_blocks,
                              hipStream_t stream, const char **name, hipEvent_t dominant_done,
                              AfterDominant after, void *user, hipStream_t tail,
                              const StackArgs *fused_replay, unsigned fused_replay_blocks)
{
    hipError_t err = hipSuccess;
    FastArgs f = fargs;
    f.in_list = nullptr;
    f.in_count = nullptr;
    f.in_capacity = 0;
    if constexpr (NS >= kZonalMinSize) {
        f.cont_list = nullptr; f.cont_state = nullptr; f.cont_count = nullptr; f.cont_region = 0; f.in_state = nullptr;
        f.in_region = f.in_regions = f.in_group = 0;
        f.pass_budget = f.round_cap = 0;
        if (WINSOR && fargs.cas_list[0]) {            // first stage of the winsorization cascade: one region per workgroup
            f.cont_list = fargs.cas_list[0];
            f.cont_state = fargs.cas_state[0];
            f.cont_count = fargs.cas_count[0];
            f.cont_region = 256;
            f.pass_budget = fargs.cas_pass[0];
            f.round_cap = fargs.cas_cap[0];
        }
        // (workgroups of 64 or 128 threads instead of 256 -- no wave waits for its workgroup's slowest at the barriers of the
        // hand-over lists -- measured the same within the noise at 32 and 128 frames, round 4)
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
        if constexpr (WINSOR) {
            // winsorization cascade (stack_fast_sigma_impl.hpp): the dominant kernel above stopped at its budget; two more
            // stages over the continuation lists, in freshly packed waves, the last one without a budget
            if (fargs.cas_list[0]) {
                FastArgs g = f;
                unsigned regions = tile_blocks, region = 256;          // of the list the stage reads
                for (int st = 1; st < fargs.cas_stages; st++) {
                    const bool last = st == fargs.cas_stages - 1;
                    const unsigned group = (unsigned)fargs.cas_group[st];
                    const unsigned blocks = (regions + group - 1) / group;
                    g.in_list = fargs.cas_list[(st - 1) & 1];
                    g.in_state = fargs.cas_state[(st - 1) & 1];
                    g.in_count = fargs.cas_count[(st - 1) & 1];
                    g.in_capacity = 0;
                    g.in_region = region;
                    g.in_regions = regions;
                    g.in_group = group;
                    g.cont_list = last ? nullptr : fargs.cas_list[st & 1];
                    g.cont_state = last ? nullptr : fargs.cas_state[st & 1];
                    g.cont_count = last ? nullptr : fargs.cas_count[st & 1];
                    g.cont_region = region * group;                   // (a workgroup cannot hand on more than it was given)
                    g.pass_budget = last ? 0 : fargs.cas_pass[st];
                    g.round_cap = last ? 0 : fargs.cas_cap[st];
                    if (args.n_frames == NS)
                        hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, true, WINSOR, true, false, true>), dim3(blocks), dim3(256), 0, stream, args, g);
                    else
                        hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, true, WINSOR, false, false, true>), dim3(blocks), dim3(256), 0, stream, args, g);
                    keep_first(err, hipGetLastError());
                    regions = blocks;
                    region = region * group;
                }
            }
        }
        if (after) after(user);
        // generic pass over the pixels the zonal waves handed over (its length
        // is only known on the device: fixed grid, grid-stride loop); chunked passes run it on
        // a stream of its own (`tail`, ordered behind the dominant kernel by the caller's callback)
        if (tail) stream = tail;
        f.in_list = fargs.gen_list;
        f.in_count = fargs.gen_count;
        f.in_capacity = fargs.gen_capacity;
        const unsigned gblocks = generic_grid(fargs.gen_hint, 256, tile_blocks < kGenericGrid ? tile_blocks : kGenericGrid);
        if constexpr (NS > 64) {
            // whole columns + prefix sums in LDS (stack_fast_mlg.hip): a clipping or winsorization round
            // is a few LDS reads instead of a pass over 128 masked registers -- this pass is pure
            // latency (a few hundred waves at most), and it sits on every pass's critical path
            const unsigned lblocks = generic_grid(fargs.gen_hint, 64, 4 * tile_blocks < 4 * kGenericGrid ? 4 * tile_blocks : 4 * kGenericGrid);
            if (!WINSOR && fused_replay)      // generic pass + first replay in one grid (stack_tail_fused.hip)
                keep_first(err, launch_stack_sigma_tail(args, f, lblocks, *fused_replay, fused_replay_blocks, stream));
            else
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
                         hipEvent_t dominant_done, AfterDominant after, void *user, hipStream_t tail,
                         const StackArgs *fused_replay, unsigned fused_replay_blocks)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    // network sizes: the frame count rounded up to the next instantiated size;
    // unused positions count as missing samples
    if (n <= 8)        return launch_pair<8, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 16)  return launch_pair<16, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 24)  return launch_pair<24, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 32)  return launch_pair<32, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 48)  return launch_pair<48, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 64)  return launch_pair<64, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 80)  return launch_pair<80, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 96)  return launch_pair<96, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else if (n <= 112) return launch_pair<112, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    else               return launch_pair<128, WINSOR>(args, fargs, blocks, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
}

hipError_t launch_stack_sigma_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                   const char **name, hipEvent_t dominant_done,
                                   bool winsor, AfterDominant after, void *user, hipStream_t tail,
                                   const StackArgs *fused_replay, unsigned fused_replay_blocks)
{
    hipError_t err = winsor ? launch_sized<true>(args, fargs, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks)
                            : launch_sized<false>(args, fargs, stream, name, dominant_done, after, user, tail, fused_replay, fused_replay_blocks);
    keep_first(err, hipGetLastError());
    return err;
}

}  // namespace nl
