// fast_common.hpp -- compile-time building blocks shared by the register-resident
// kernels (stack_fast.hip, stack_fast_ml.hip): static loops, the odd-even merge
// sorting network, per-lane picks from a register column.
#pragma once
#include <type_traits>
#include <utility>

#include "stack_kernels.h"

namespace nl {

constexpr float kU = 5.9604644775390625e-8f;   // 2^-24, fp32 unit roundoff
constexpr int kZone = 8;      // sorted positions per side that may be clipped in the zonal path
constexpr int kPadMax = 8;    // missing samples (NaN) a lane may have in the zonal path
constexpr unsigned kGenericGrid = 2048;   // workgroups of the generic pass over the hand-over list
// The list's length is only known on the device, so the generic pass is a fixed grid with a grid-stride loop --
// but thousands of empty workgroups (48 KiB of LDS each in the LDS generic pass: three per CU) take longer to
// drain than a short list takes to process.  The length the last finished pass reported sizes the grid: half
// as many workgroups again as it needed, at least 64.
inline unsigned generic_grid(unsigned hint, unsigned pixels_per_wg, unsigned max_grid)
{
    if (hint == 0) return max_grid;
    const unsigned need = (hint - 1 + pixels_per_wg - 1) / pixels_per_wg;
    const unsigned g = need + need / 2 + 64;
    return g < max_grid ? g : max_grid;
}

// A generic pass appends to the exact-replay list while the first replay may already be reading it:
// before its first append every workgroup makes sure the list's length as the dominant kernel left it
// is on record (StackArgs::list_snap; the first to look -- of this pass or of the replay -- wins).
__device__ __forceinline__ void snapshot_fb_list(const FastArgs &q)
{
    // (a plain look first: thousands of workgroups hitting one address with atomics take 0.5 ms)
    // Ordering: the CAS is a RETURNING atomic whose value is consumed, so it has been performed at L2
    // (device scope) before this thread goes on; the fence orders it before the barrier, and no wave of
    // this workgroup appends (atomicAdd on fb_count) before the barrier.  The length itself is read with
    // an atomic load: appends of OTHER workgroups may be in flight, but each of them has recorded the
    // snapshot before its first append, so whoever wins the CAS read a count no append had touched.
    if (q.fb_snap && threadIdx.x == 0 && __atomic_load_n(q.fb_snap, __ATOMIC_RELAXED) == 0u) {
        const unsigned len = __atomic_load_n(q.fb_count, __ATOMIC_RELAXED);
        const unsigned prev = atomicCAS(q.fb_snap, 0u, len + 1u);
        if (prev == 0xffffffffu) __builtin_trap();           // (never: keeps the returning form)
        __threadfence();
    }
    __syncthreads();
}

// ---- fused pass protocol (StackArgs::final, see stack_kernels.h and nlstack_api.hip) ----
// dominant kernel, first workgroup: zero this pass's totals and the scratch set of the NEXT pass (everything
// that used either is stream-ordered before this kernel)
__device__ __forceinline__ void fused_prologue_dominant(const StackArgs &p)
{
    if (p.zero_next && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < kScratchWords; i += blockDim.x) p.zero_next[i] = 0ull;
        if (threadIdx.x < 4) p.final[threadIdx.x] = 0ull;
    }
}
// generic pass, first wave of the first workgroup: the dominant kernel's sharded counts -> totals
// (`block`: this workgroup's index within the generic pass -- its position in the grid unless the pass shares a grid,
// stack_tail_fused.hip)
__device__ __forceinline__ void fused_collect_slots(const StackArgs &p, unsigned block)
{
    if (p.final && block == 0 && threadIdx.x < 64) {
        unsigned long long lo = 0, hi = 0;
        for (int i = threadIdx.x; i < kClipSlots; i += 64) {
            lo += p.partial[2 * (size_t)i];
            hi += p.partial[2 * (size_t)i + 1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo += ((unsigned long long)(unsigned)__shfl_xor((int)(lo >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)lo, o, 64);
            hi += ((unsigned long long)(unsigned)__shfl_xor((int)(hi >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)hi, o, 64);
        }
        if (threadIdx.x == 0) {
            if (lo) atomicAdd(p.final + 0, lo);
            if (hi) atomicAdd(p.final + 1, hi);
        }
    }
}
// where a kernel that runs AFTER the dominant one adds its clip counts: the totals, or (plain protocol) a shard
__device__ __forceinline__ unsigned long long *clip_slot(const StackArgs &p, unsigned block)
{
    return p.final ? p.final : p.partial + 2 * (size_t)(block % kClipSlots);
}

// compile-time loops: every index is a constant, so register columns never
// fall back to scratch memory (pragma unroll gives up on the large networks)
template <int B, int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void static_range(F &&f)
{
    if constexpr (E > B) static_for_impl<B>(std::make_integer_sequence<int, E - B>{}, static_cast<F &&>(f));
}
// same, but the instruction scheduler may not move code across chunk
// boundaries: keeps the live ranges of per-element temporaries (lane masks,
// scalar addresses, differences) short in these very long basic blocks
template <int B, int E, int CH, class F>
__device__ __forceinline__ void static_chunks(F &&f)
{
    if constexpr (E > B) {
        constexpr int M = (B + CH < E) ? B + CH : E;
        static_for_impl<B>(std::make_integer_sequence<int, M - B>{}, f);
        __builtin_amdgcn_sched_barrier(0);
        static_chunks<M, E, CH>(static_cast<F &&>(f));
    }
}
#define NL_INL __attribute__((always_inline))

// ---- Batcher odd-even merge sort, generated at compile time -----------------
// All comparators put the minimum at the lower index, so comparators touching
// an index >= NS can simply be dropped: the network sorts NS elements for any
// NS (not only powers of two).
struct CePair { short lo, hi; };

template <int NS>
struct OemNetwork {
    static constexpr int pow2()
    {
        int p = 1;
        while (p < NS) p <<= 1;
        return p;
    }
    static constexpr int count()
    {
        int c = 0;
        const int P2 = pow2();
        for (int p = 1; p < P2; p <<= 1)
            for (int k = p; k >= 1; k >>= 1)
                for (int j = k % p; j + k < P2; j += 2 * k)
                    for (int i = 0; i < k; i++)
                        if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && (i + j + k) < NS) c++;
        return c;
    }
    static constexpr int kCount = count();
    struct Table { CePair e[kCount > 0 ? kCount : 1]; };
    static constexpr Table make()
    {
        Table t{};
        int c = 0;
        const int P2 = pow2();
        for (int p = 1; p < P2; p <<= 1)
            for (int k = p; k >= 1; k >>= 1)
                for (int j = k % p; j + k < P2; j += 2 * k)
                    for (int i = 0; i < k; i++)
                        if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && (i + j + k) < NS) {
                            t.e[c].lo = (short)(i + j);
                            t.e[c].hi = (short)(i + j + k);
                            c++;
                        }
        return t;
    }
    static constexpr Table kTable = make();
};

// ---- the networks as executed: 2- and 3-input operations -------------------
// v_min_f32 / v_max_f32 and v_min3 / v_med3 / v_max3 all issue at the same (half) rate on gfx950,
// so a network costs its instruction count.  tools/gen_sort_tables.py rewrites each network
// (the comparator lists above / ZonalNetwork below) into operations on numbered slots, inlining
// an intermediate min or max into the comparator that consumes it wherever the 0-1 principle
// proves  CE(min(x,y), z) == {min3(x,y,z), med3(x,y,z)}  (resp. {med3, max3} for max(x,y)):
// about a quarter fewer instructions.  kOut[k] = slot holding rank k at the end.
struct FusedOp { unsigned char kind; short dst, a, b, c; };   // kind: 0 min, 1 max, 2 min3, 3 med3, 4 max3
template <int NS, int E0, int E1, int E2, int E3>
struct FusedNet;                                               // specialisations: sort_tables.inc
#include "sort_tables.inc"

// ASM: the 2-input and min3 / max3 operations are written as the instructions themselves --
// fminf / fmaxf put a canonicalising v_max_f32 x, x, x in front of every value that comes
// straight from memory (the column holds no NaN here, so there is nothing to quiet).  Leaving
// them to the compiler instead lets it schedule more freely, which the MAD kernel (two sorts at
// 168 registers) prefers.
// (NA: the array may be longer than the network -- the network then orders v[0 .. NS) and leaves the rest)
template <class Net, int NS, bool ASM = true, int NA = NS>
__device__ __forceinline__ void run_network(float (&v)[NA])
{
    static_assert(NS <= NA, "network larger than the column");
    float w[Net::kSlots];
    static_range<0, NS>([&](auto K) NL_INL { w[decltype(K)::value] = v[decltype(K)::value]; });
    static_chunks<0, Net::kCount, 128>([&](auto I) NL_INL {
        constexpr FusedOp op = Net::kOps[decltype(I)::value];
        float r;
        if constexpr (op.kind == 3)      r = __builtin_amdgcn_fmed3f(w[op.a], w[op.b], w[op.c]);
        else if constexpr (!ASM && op.kind == 0) r = fminf(w[op.a], w[op.b]);
        else if constexpr (!ASM && op.kind == 1) r = fmaxf(w[op.a], w[op.b]);
        else if constexpr (!ASM && op.kind == 2) r = fminf(fminf(w[op.a], w[op.b]), w[op.c]);
        else if constexpr (!ASM && op.kind == 4) r = fmaxf(fmaxf(w[op.a], w[op.b]), w[op.c]);
        else if constexpr (op.kind == 0) asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(w[op.a]), "v"(w[op.b]));
        else if constexpr (op.kind == 1) asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(w[op.a]), "v"(w[op.b]));
        else if constexpr (op.kind == 2) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(w[op.a]), "v"(w[op.b]), "v"(w[op.c]));
        else                             asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(w[op.a]), "v"(w[op.b]), "v"(w[op.c]));
        w[op.dst] = r;
    });
    static_range<0, NS>([&](auto K) NL_INL { v[decltype(K)::value] = w[Net::kOut[decltype(K)::value]]; });
}

template <int NS, bool ASM = true, int NA = NS>
__device__ __forceinline__ void sort_network(float (&v)[NA])
{
    using Net = FusedNet<NS, 0, 0, 0, 0>;
    static_assert(Net::kComparators == OemNetwork<NS>::kCount, "sort_tables.inc does not match OemNetwork");
    run_network<Net, NS, ASM, NA>(v);
}

// The zonal kernels need exact ranks only at the ends (clip zones) and around
// the middle (median window) of the sorted column: positions [0,E0), [E1,E2)
// and [E3,NS).  The two stretches in between are only ever summed, so they
// merely have to hold the right SET of samples.  Walking the network backwards,
// a comparator is dropped if both its wires end in the same such stretch and
// no later kept comparator touches either wire: it could only swap two values
// inside the stretch.  (1313 of 1471 comparators remain for NS = 128.)
template <int NS, int E0, int E1, int E2, int E3>
struct ZonalNetwork {
    using Full = OemNetwork<NS>;
    static constexpr int stretch(int i) { return i < E0 ? 0 : (i < E1 ? 1 : (i < E2 ? 0 : (i < E3 ? 2 : 0))); }
    struct Table { CePair e[Full::kCount > 0 ? Full::kCount : 1]; int n; };
    static constexpr Table make()
    {
        Table t{};
        bool touched[NS] = {};
        bool keep[Full::kCount > 0 ? Full::kCount : 1] = {};
        for (int c = Full::kCount - 1; c >= 0; c--) {
            const CePair ce = Full::kTable.e[c];
            const int sl = stretch(ce.lo), sh = stretch(ce.hi);
            if (sl == sh && sl != 0 && !touched[ce.lo] && !touched[ce.hi]) continue;
            keep[c] = true;
            touched[ce.lo] = true;
            touched[ce.hi] = true;
        }
        int n = 0;
        for (int c = 0; c < Full::kCount; c++)
            if (keep[c]) t.e[n++] = Full::kTable.e[c];
        t.n = n;
        return t;
    }
    static constexpr Table kTable = make();
    static constexpr int kCount = kTable.n;
};

template <bool ASM>
struct FullSortT {
    template <int NS>
    static __device__ __forceinline__ void apply(float (&v)[NS]) { sort_network<NS, ASM>(v); }
};
using FullSort = FullSortT<true>;
template <int E0, int E1, int E2, int E3>
struct ZonalSort {
    template <int NS>
    static __device__ __forceinline__ void apply(float (&v)[NS])
    {
        using Net = FusedNet<NS, E0, E1, E2, E3>;
        static_assert(Net::kComparators == ZonalNetwork<NS, E0, E1, E2, E3>::kCount,
                      "sort_tables.inc does not match ZonalNetwork");
        run_network<Net, NS>(v);
    }
};

// the compiler must not share the 'rank in [a,b)' masks between passes: 128
// live lane masks would spill the SGPR file
__device__ __forceinline__ int opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// value at a per-lane position idx, known to lie in [B, E).  The index is
// re-materialised every 8 candidates so that at most 8 compare masks are live.
template <int B, int E, int NS>
__device__ __forceinline__ float pick(const float (&v)[NS], int idx)
{
    float r = v[B];
    static_range<B + 1, E>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        if constexpr (((k - B) & 7) == 0) idx = opaque(idx);
        r = (idx == k) ? v[k] : r;
    });
    return r;
}

// v[idx] and v[idx-1] with one compare per candidate (the two ranks a median reads);
// idx in [B, E); lower is only meaningful for idx > B
template <int B, int E, int NS>
__device__ __forceinline__ void pick_pair(const float (&v)[NS], int idx, float &lower, float &upper)
{
    float up = v[B], lo = v[B];
    static_range<B + 1, E>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        if constexpr (((k - B) & 7) == 0) idx = opaque(idx);
        const bool hit = idx == k;
        up = hit ? v[k] : up;
        lo = hit ? v[k - 1] : lo;
    });
    upper = up;
    lower = lo;
}

// min / max as the instructions themselves: fminf / fmaxf put a canonicalising v_max_f32 x, x, x
// in front of every operand that comes out of the (asm) sorting network.  No NaN reaches these.
__device__ __forceinline__ float max_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min_raw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// NaN -> +Inf, everything else unchanged: IEEE minNum(NaN, Inf) = Inf.  Written
// as the instruction itself so that it stays ONE VALU op without a lane mask.
__device__ __forceinline__ float nan_to_inf(float x)
{
    float y;
    asm("v_min_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(__builtin_inff()));
    return y;
}

// Gather one pixel's samples into registers and sort them ascending; returns
// the number of valid samples n (they occupy v[0..n), +Inf above).
// NaN = no data (stack.go:380-387): NaNs and unused positions (k >= N) become
// +Inf and sort last.  All loads are issued first (independent, 256 B per wave
// each); the frame pointer advances by one frame per position and stops at the
// last frame, so unused positions re-read a valid address.
// GAP: the caller guarantees N > NS - GAP (distance to its next smaller network size).
// SORT: FullSort, or ZonalSort<...> where only part of the order is needed.
// PADDED = false: the caller knows N == NS (no unused positions).
template <int NS, int GAP = 16, class SORT = FullSort, bool NT = false, bool PADDED = true>
__device__ __forceinline__ int gather_sorted(const float *frames, int64_t stride, int N,
                                             unsigned boff, float (&v)[NS], int lo_pads = 0)
{
    // Buffer loads: the address is (scalar descriptor base) + (scalar offset) +
    // (one per-lane byte offset), so the 128 loads need neither per-load VGPR
    // address pairs nor branches.  Four frames share a descriptor; the frame
    // index is clamped to the last frame (positions k >= N re-read it and are
    // turned into missing samples below).
    const int64_t frame_bytes = stride * (int64_t)sizeof(float);
    const int last = N - 1;
    // NT: cache policy nt (aux = 2) for kernels that read every frame byte exactly once per pass --
    // the lines need not stay in L2 / MALL: the median kernel gains 6 % (80 % of the 8 TB/s peak).
    // Not for the MAD kernel (its second read of the column hits the MALL) nor for the multi-lane
    // gathers (neighbouring waves share 128-byte lines through L2): both measured slower with nt.
    static_chunks<0, NS / 4, 4>([&](auto C) NL_INL {
        constexpr int c0 = 4 * decltype(C)::value;
        const int f0 = min(c0, last);
        const char *gb = reinterpret_cast<const char *>(frames) + (int64_t)f0 * frame_bytes;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gb), 0, -1, 0x00020000);
        static_range<0, 4>([&](auto U) NL_INL {
            constexpr int k = c0 + decltype(U)::value;
            const int soff = (min(k, last) - f0) * (int)frame_bytes;
            v[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)boff, soff, NT ? 2 : 0));
        });
    });
    // Unused positions k >= N (N > NS - GAP by the choice of NS) hold no frame: +Inf (they sort last), and they stay
    // out of the finiteness test below -- otherwise every wave of a stack whose frame count is not a network size
    // would take the NaN count (about 4 instructions per position) for the sake of its own padding.
    // Clean waves (no lane holds a NaN or an infinite sample -- everything but the aligned frames' borders) skip the
    // NaN count: a plain fp32 sum of the column is finite iff every sample is.  (A sum that overflows only costs the count.)
    constexpr int P0 = !PADDED ? NS : (NS > GAP ? NS - GAP : 0);
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    static_range<P0, NS>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        const int pad = (N - 1 - k) >> 31;                        // scalar: N is uniform
        // lo_pads of the unused positions become -Inf instead (they sort FIRST): the zonal kernels of a stack with
        // many unused positions start with their low pointer behind them, see stack_fast_sigma_impl.hpp
        const int inf = 0x7f800000 | (((k - N - lo_pads) >> 31) & (int)0x80000000);
        t0 += __int_as_float(__float_as_int(v[k]) & ~pad);
        v[k] = __int_as_float((__float_as_int(v[k]) & ~pad) | (inf & pad));
    });
    static_chunks<0, P0 / 4, 8>([&](auto K) NL_INL {
        constexpr int k = 4 * decltype(K)::value;
        t0 += v[k]; t1 += v[k + 1]; t2 += v[k + 2]; t3 += v[k + 3];
    });
    static_range<P0 / 4 * 4, P0>([&](auto K) NL_INL { t1 += v[decltype(K)::value]; });
    const float total = (t0 + t1) + (t2 + t3);
    int nan_cnt = 0;
    if (__any(!(__builtin_fabsf(total) < __builtin_inff()))) {
        // NaN <=> (bits & 0x7fffffff) > 0x7f800000; counted with integer arithmetic
        // (a compare would park a lane mask in SGPRs per element), then NaN -> +Inf in
        // place (tied asm operand: the column keeps its registers across the branch)
        static_chunks<0, NS, 8>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            nan_cnt = opaque(nan_cnt - ((0x7f800000 - (__float_as_int(v[k]) & 0x7fffffff)) >> 31));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[k]) : "v"(__builtin_inff()));
        });
    }
    SORT::template apply<NS>(v);
    return min(N, NS) - nan_cnt;                              // (the padding is +Inf, not NaN: not in nan_cnt)
}

// ---- winsorization loop of StackWinsorSigma (stack.go:646-672) on intervals ------
// The reference repeats { clamp a copy to median -/+ 1.5*std; std = 1.134 *
// stddev(copy) } until no sample moved or std changed by <= 0.05 %.  Its std is
// an order-dependent fp32 sum, so the register-resident kernels carry an
// interval [w_lo, w_hi] that contains it.  The clamp bounds become intervals
// too; the variance of the clamped copy is monotone in the clamp (narrowing a
// clamp never increases a pairwise distance), so evaluating it at the tightest
// clamp (Lp, Hm) and at the loosest (Lm, Hp) brackets it.  Where an exit test is
// undecidable the reference EITHER left the loop with a value in the current
// interval OR went on: we go on and keep the hull of every value it may have
// left with; the clip step must then be unambiguous over that hull.
struct WinsorInterval {
    float w_lo, w_hi;              // the reference's current std lies in here
    float Lm, Lp, Hm, Hp;          // effective clamp: low bound in [Lm, Lp], high bound in [Hm, Hp]
    float hull_lo, hull_hi;        // values the reference may have left the loop with
    int guard, guard_max;
    bool ch_sure, ch_none;         // "changed > 0" certain / "changed == 0" certain, this round

    // max_rounds: a pixel that is still inside the loop after that many rounds goes to the exact replay
    __device__ __forceinline__ void start(float s_min, float s_max, int max_rounds = 100)
    {
        guard_max = max_rounds;
        w_lo = s_min; w_hi = s_max;
        Lm = Lp = -__builtin_inff();           // running max of the low bounds
        Hm = Hp = __builtin_inff();            // running min of the high bounds
        hull_lo = __builtin_inff(); hull_hi = -__builtin_inff();
        guard = 0;
    }
    // bounds of this round (stack.go:650-651) and the changed == 0 test (:652-662):
    // xmin / xmax = smallest / largest surviving sample
    __device__ __forceinline__ void next_clamp(float median, float xmin, float xmax)
    {
        const float tA = __fmul_rn(1.5f, w_lo), tB = __fmul_rn(1.5f, w_hi);
        const float lo_m = __fsub_rn(median, tB), lo_p = __fsub_rn(median, tA);   // low bound in [lo_m, lo_p]
        const float hi_m = __fadd_rn(median, tA), hi_p = __fadd_rn(median, tB);   // high bound in [hi_m, hi_p]
        const float wmin_m = fmaxf(xmin, Lm), wmin_p = fmaxf(xmin, Lp);           // smallest value of the copy
        const float wmax_m = fminf(xmax, Hm), wmax_p = fminf(xmax, Hp);           // largest
        ch_sure = (wmin_p < lo_m) || (wmax_m > hi_p);
        ch_none = (wmin_m >= lo_p) && (wmax_p <= hi_m);
        Lm = fmaxf(Lm, lo_m); Lp = fmaxf(Lp, lo_p);
        Hm = fminf(Hm, hi_m); Hp = fminf(Hp, hi_p);
    }
    // var_t +- err_t: variance of the copy at the tightest clamp, var_l +- err_l at the
    // loosest; eps_r, e_m: rounding of the reference's own MeanStdDev.
    // Clears `inner` when the loop is certainly over, sets `bail` if the pixel has to be
    // replayed exactly.
    __device__ __forceinline__ void finish_round(float var_t, float err_t, float var_l, float err_l,
                                                 float eps_r, float e_m, bool shape_ok, bool &inner, bool &bail)
    {
        const float w_up = var_l + err_l;
        const float w_dn = fmaxf(var_t - err_t, 0.0f);
        // hardware sqrt (1 ulp, flushes denormals): the 4u margins and the absolute term cover it
        const float r_hi = __builtin_amdgcn_sqrtf(w_up + w_up * eps_r + e_m * e_m) * (1.0f + 4.0f * kU) + 4.0e-19f;
        const float r_lo = __builtin_amdgcn_sqrtf(fmaxf(w_dn - w_dn * eps_r, 0.0f)) * (1.0f - 4.0f * kU);
        const float n_lo = __fmul_rn(1.134f, r_lo), n_hi = __fmul_rn(1.134f, r_hi);
        // factor = |new - old| / old <= 0.0005 (stack.go:668-669) without the division:
        // fl(x/y) <= t follows from x <= y*t*(1-4u), fl(x/y) > t from x > y*t*(1+4u)
        const float dmin = __fsub_rn(n_lo, w_hi), dmax = __fsub_rn(n_hi, w_lo);
        const float amin = (dmin <= 0.0f && dmax >= 0.0f) ? 0.0f : fminf(fabsf(dmin), fabsf(dmax));
        const float amx = fmaxf(fabsf(dmin), fabsf(dmax));
        const bool stop_sure = w_lo > 0.0f && amx <= w_lo * (0.0005f * (1.0f - 4.0f * kU));
        const bool go_sure = amin > w_hi * (0.0005f * (1.0f + 4.0f * kU));
        if (inner) {
            w_lo = n_lo; w_hi = n_hi;
            const bool may_stop = !ch_sure || !go_sure;
            const bool must_stop = ch_none || stop_sure;          // implies may_stop
            if (may_stop) { hull_lo = fminf(hull_lo, w_lo); hull_hi = fmaxf(hull_hi, w_hi); }
            if (!shape_ok || !(w_hi < 3.0e38f) || ++guard > guard_max) { bail = true; inner = false; }
            else if (must_stop) inner = false;
        }
    }
};

}  // namespace nl
