// fast_ml_common.hpp -- building blocks of the multi-lane kernels (2 or 4 adjacent lanes
// share one pixel, 128 samples per lane): DPP quad permutes and reductions, the cross-lane
// bitonic merge, rank lookups, and the gather that ends with lane r holding the global sorted
// ranks [128r, 128r+128).  Used by stack_fast_ml.hip (sigma / winsor / median) and
// stack_linfit.hip (linear fit beyond 128 frames).
#pragma once
#include "fast_common.hpp"

namespace nl {

constexpr int kMlNS = 128;     // samples per lane

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x)
{
    return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true);
}
constexpr int kSwap1 = 0xB1;   // quad_perm [1,0,3,2]: partner = lane ^ 1
constexpr int kSwap2 = 0x4E;   // quad_perm [2,3,0,1]: partner = lane ^ 2
constexpr int kMirror = 0x1B;  // quad_perm [3,2,1,0]: partner = 3 - lane

// sums / ors over the LPP lanes of a pixel; every lane gets the same bits
template <int LPP>
__device__ __forceinline__ float quad_sum(float x)
{
    x = x + dpp_f<kSwap1>(x);
    if constexpr (LPP == 4) x = x + dpp_f<kSwap2>(x);
    return x;
}
template <int LPP>
__device__ __forceinline__ int quad_sum(int x)
{
    x = x + dpp_i<kSwap1>(x);
    if constexpr (LPP == 4) x = x + dpp_i<kSwap2>(x);
    return x;
}
template <int LPP>
__device__ __forceinline__ int quad_or(int x)
{
    x = x | dpp_i<kSwap1>(x);
    if constexpr (LPP == 4) x = x | dpp_i<kSwap2>(x);
    return x;
}

// the pixel's lane `R`, in every lane of the pixel
template <int LPP, int R>
__device__ __forceinline__ int quad_bcast(int x)
{
    if constexpr (LPP == 2) return dpp_i<R == 0 ? 0xA0 : 0xF5>(x);            // quad_perm [0,0,2,2] / [1,1,3,3]
    else return dpp_i<R == 0 ? 0x00 : (R == 1 ? 0x55 : (R == 2 ? 0xAA : 0xFF))>(x);
}

// in-lane half-cleaners of a bitonic merge: distances NS/2 ... 1, ascending
template <int NS, int D, int CH = 32>
__device__ __forceinline__ void half_clean(float (&v)[NS])
{
    if constexpr (D >= 1) {
        static_chunks<0, NS / 2, CH>([&](auto T) NL_INL {
            constexpr int t = decltype(T)::value;
            constexpr int i = ((t & ~(D - 1)) << 1) | (t & (D - 1));
            constexpr int l = i | D;
            const float lo = fminf(v[i], v[l]);
            const float hi = fmaxf(v[i], v[l]);
            v[i] = lo;
            v[l] = hi;
        });
        half_clean<NS, (D >> 1), CH>(v);
    }
}

// The last merge of the zonal kernels: only the KEEP lowest and KEEP highest
// positions of the lane must end up sorted (clip zones, median window); all
// other positions are only ever summed, so they merely have to hold the right
// SET.  A half-cleaner block that cannot reach either end is skipped:
// 224 instead of 448 compare-exchanges for NS = 128, KEEP = 16.
template <int NS, int D, int KEEP, int CH = 32>
__device__ __forceinline__ void half_clean_ends(float (&v)[NS])
{
    if constexpr (D >= 1) {
        static_chunks<0, NS / 2, CH>([&](auto T) NL_INL {
            constexpr int t = decltype(T)::value;
            constexpr int i = ((t & ~(D - 1)) << 1) | (t & (D - 1));
            constexpr int l = i | D;
            constexpr int blk = i & ~(2 * D - 1);                 // this comparator's block [blk, blk + 2D)
            if constexpr (blk < KEEP || blk + 2 * D > NS - KEEP) {
                const float lo = fminf(v[i], v[l]);
                const float hi = fmaxf(v[i], v[l]);
                v[i] = lo;
                v[l] = hi;
            }
        });
        half_clean_ends<NS, (D >> 1), KEEP, CH>(v);
    }
}

// cross-lane stage against the partner selected by CTRL; MIRROR: element i
// meets the partner's element NS-1-i (first stage of merging two ascending
// runs), else element i meets element i (half-cleaner at lane distance)
template <int NS, int CTRL, bool MIRROR>
__device__ __forceinline__ void cross_stage(float (&v)[NS], bool keep_min)
{
    // min(x, y) = med3(x, y, -Inf), max(x, y) = med3(x, y, +Inf): one VALU op per
    // element whichever side of the exchange the lane is on (no NaN can occur here)
    const float side = keep_min ? -__builtin_inff() : __builtin_inff();
    if constexpr (MIRROR) {
        static_chunks<0, NS / 2, 16>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            constexpr int j = NS - 1 - i;
            const float pj = dpp_f<CTRL>(v[j]);
            const float pi = dpp_f<CTRL>(v[i]);
            v[i] = __builtin_amdgcn_fmed3f(v[i], pj, side);
            v[j] = __builtin_amdgcn_fmed3f(v[j], pi, side);
        });
    } else {
        static_chunks<0, NS, 32>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            const float pv = dpp_f<CTRL>(v[i]);
            v[i] = __builtin_amdgcn_fmed3f(v[i], pv, side);
        });
    }
}

// value at global sorted rank g (all lanes of the pixel receive it).  TOPW/BOTW
// restrict the lookup to the last TOPW registers of lane `lo_role` and the first
// BOTW registers of lane lo_role+1 (the only places g can be in the zonal
// passes); TOPW = BOTW = NS searches everything.
template <int LPP, int NS, int TOPW, int BOTW>
__device__ __forceinline__ float pick_rank(const float (&v)[NS], int g, int role, int lo_role)
{
    const int local = g - role * NS;
    float r = 0.0f;
    bool hit = false;
    if constexpr (TOPW >= NS) {
        hit = (unsigned)local < (unsigned)NS;
        r = pick<0, NS>(v, local);
    } else {
        const bool top = role == lo_role && local >= NS - TOPW && local < NS;
        const bool bot = role == lo_role + 1 && local >= 0 && local < BOTW;
        const float rt = pick<NS - TOPW, NS>(v, local);
        const float rb = pick<0, BOTW>(v, local);
        hit = top || bot;
        r = top ? rt : rb;
    }
    return __int_as_float(quad_or<LPP>(hit ? __float_as_int(r) : 0));
}

// ZONAL / generic exactly as in stack_fast.hip; LPP lanes per pixel.
// Sort every lane's column and merge the LPP runs of a pixel: afterwards lane r holds the global
// ranks [r*NS, r*NS+NS).  ENDS_ONLY: the last merge orders only the KEEP lowest / highest ranks
// of every lane (see half_clean_ends).
// NSL: positions [NSL, NS) of every lane are known to hold +Inf (frames the stack does not have: the caller's
// frame-count class), so the in-lane sort is the network of NSL positions -- 1 100 instead of 2 184 operations
// for a 300-frame stack on four lanes
template <int LPP, int NS, bool ENDS_ONLY, int KEEP = 16, int CH = 32, int NSL = NS>
__device__ __forceinline__ void ml_sort_merge(float (&v)[NS], int role)
{
    sort_network<NSL, true, NS>(v);
    // ---- merge the LPP sorted runs: lane r ends up with ranks [r*NS, r*NS+NS) ----
    // (zonal: the final half-cleaners only order the ends of each lane, see half_clean_ends)
    static_assert(kZone + kPadMax <= KEEP, "zones must lie inside the sorted ends");
    cross_stage<NS, kSwap1, true>(v, (role & 1) == 0);
    // the in-lane half-cleaners run as 2-/3-input operations too (FusedBitonic, sort_tables.inc: the
    // 0-1 analysis over bitonic inputs fuses 30 % of the instructions away)
    if constexpr (NS == 128 && (KEEP == 16 || KEEP == 32)) {
        if constexpr (ENDS_ONLY && LPP == 2) run_network<FusedBitonic<NS, KEEP>, NS>(v);
        else                             run_network<FusedBitonic<NS, 0>, NS>(v);
        if constexpr (LPP == 4) {
            cross_stage<NS, kMirror, true>(v, role < 2);
            cross_stage<NS, kSwap1, false>(v, (role & 1) == 0);
            if constexpr (ENDS_ONLY) run_network<FusedBitonic<NS, KEEP>, NS>(v);
            else                 run_network<FusedBitonic<NS, 0>, NS>(v);
        }
    } else {
        if constexpr (ENDS_ONLY && LPP == 2) half_clean_ends<NS, NS / 2, KEEP, CH>(v);
        else                             half_clean<NS, NS / 2, CH>(v);
        if constexpr (LPP == 4) {
            cross_stage<NS, kMirror, true>(v, role < 2);
            cross_stage<NS, kSwap1, false>(v, (role & 1) == 0);
            if constexpr (ENDS_ONLY) half_clean_ends<NS, NS / 2, KEEP, CH>(v);
            else                 half_clean<NS, NS / 2, CH>(v);
        }
    }
}

// Gather one pixel's frames into the LPP lanes that share it (128 per lane), sort every
// lane's column and merge the runs: afterwards lane r holds global ranks [r*NS, r*NS+NS)
// (+Inf for missing samples at the top).  Returns the number of valid samples of the pixel.
// ENDS_ONLY: the last merge orders only the KEEP lowest / highest ranks of every lane.
// The gather alone: the lane's samples as loaded, NaN and missing frames as +Inf; returns the number of valid
// samples of the PIXEL (all its lanes).
// KPAD0 / NLOAD (frame-count classes, stack_fast_mlz_impl.hpp): positions below KPAD0 always hold a frame, positions from
// NLOAD on never do -- they are +Inf without a load -- and only the few in between depend on the frame count: those become
// +Inf directly and stay out of the finiteness test, so that a wave without NaN samples skips the NaN count although its
// lanes are not full (with the defaults every lane of a stack that does not fill its lanes takes the 640-instruction
// count in every wave).
// NT: cache policy nt for the loads -- only where a wave reads WHOLE 128-byte lines of a frame and nobody reads them again:
// two lanes per pixel (32 pixels per wave).  Measured on the padded stride: median 256 frames
// 4.11 -> 3.84 ms, sigma 256 frames 4.62 -> 4.55; four lanes per pixel (64 bytes per frame and wave, the neighbouring wave
// takes the other half of the line through L1 / L2) LOSE 5 % with it, and so does the MAD kernel (+ 14 %: it reads the
// column a second time out of the MALL).
template <int LPP, int NS, int KPAD0 = NS / 2, int NLOAD = NS, bool NT = false>
__device__ __forceinline__ int ml_gather_raw(const float *frames, int64_t stride, int N, bool on, int64_t pix,
                                             int role, float (&v)[NS])
{
    static_assert(KPAD0 <= NLOAD && NLOAD <= NS, "positions");
    constexpr bool CLASS = !(KPAD0 == NS / 2 && NLOAD == NS);
    int nan_cnt = 0;
    int spad = NS - NLOAD;                                   // positions of this lane without a frame
    {
        // Frames are dealt round-robin: lane role r takes frames r, r+LPP, ...
        // (any split works, the column is sorted afterwards).  Buffer loads: one
        // scalar descriptor per register index k covering frames k*LPP .. k*LPP+LPP-1,
        // per-lane byte offset = pixel + role * frame.  The descriptor's size is
        // cut at the last existing frame, so a lane whose frame k*LPP+role does
        // not exist reads out of range -- the hardware returns 0 without touching
        // memory -- and the position is marked missing below.  No per-lane
        // addresses, no branches; descriptors are scalar work.
        int frame_bytes = (int)(stride * (int64_t)sizeof(float));           // LPP*frame_bytes < 2^31 (dispatch)
        // opaque per trip: otherwise the 128 descriptors are hoisted out of the
        // item loop as loop invariants and spilled
        asm volatile("" : "+s"(frame_bytes));
        const int voff = (int)((unsigned)(on ? pix : 0) * 4u) + role * frame_bytes;
        static_chunks<0, NLOAD, 4>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            const int avail = min(max(N - k * LPP, 0), LPP);                // frames this descriptor covers
            const char *gb = reinterpret_cast<const char *>(frames) + (int64_t)(k * LPP) * frame_bytes;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gb), 0, avail * frame_bytes, 0x00020000);
            v[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, NT ? 2 : 0));
        });
        static_range<NLOAD, NS>([&](auto K) NL_INL { v[decltype(K)::value] = __builtin_inff(); });
        int lastp = opaque(N - 1) - role;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        if constexpr (CLASS) {
            // frame k*LPP+role >= N: no frame -> +Inf, counted, kept out of the sum below
            float tp = 0.0f;
            static_range<KPAD0, NLOAD>([&](auto K) NL_INL {
                constexpr int k = decltype(K)::value;
                const int pad = (lastp - k * LPP) >> 31;                       // all ones: no frame
                tp += __int_as_float(__float_as_int(v[k]) & ~pad);
                v[k] = __int_as_float((__float_as_int(v[k]) & ~pad) | (0x7f800000 & pad));
                spad -= pad;
            });
            static_chunks<0, KPAD0 / 4, 8>([&](auto K) NL_INL {
                constexpr int k = 4 * decltype(K)::value;
                t0 += v[k]; t1 += v[k + 1]; t2 += v[k + 2]; t3 += v[k + 3];
            });
            static_range<KPAD0 / 4 * 4, KPAD0>([&](auto K) NL_INL { t0 += v[decltype(K)::value]; });
            t0 += tp;
        } else {
            // frame k*LPP+role >= N: missing (NaN).  Only k >= KPAD0 can be affected:
            // this kernel is used for N > NT/2.
            static_chunks<KPAD0, NS, 8>([&](auto K) NL_INL {
                constexpr int k = decltype(K)::value;
                if constexpr ((k & 7) == 0) lastp = opaque(lastp);
                const int pad = (lastp - k * LPP) >> 31;                           // all ones -> NaN
                v[k] = __int_as_float(__float_as_int(v[k]) | pad);
            });
            // clean waves skip the NaN count (see gather_sorted in fast_common.hpp)
            static_chunks<0, NS / 4, 8>([&](auto K) NL_INL {
                constexpr int k = 4 * decltype(K)::value;
                t0 += v[k]; t1 += v[k + 1]; t2 += v[k + 2]; t3 += v[k + 3];
            });
        }
        const float total = (t0 + t1) + (t2 + t3);
        if (__any(!(__builtin_fabsf(total) < __builtin_inff()))) {
            static_chunks<0, NLOAD, 8>([&](auto K) NL_INL {
                constexpr int k = decltype(K)::value;
                nan_cnt = opaque(nan_cnt - ((0x7f800000 - (__float_as_int(v[k]) & 0x7fffffff)) >> 31));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[k]) : "v"(__builtin_inff()));   // NaN -> +Inf in place
            });
        }
    }
    return quad_sum<LPP>(NS - nan_cnt - spad);
}

// The gather of a frame-count class that fills its lanes (NLOAD == NS, KPAD0 >= NS / 2) together with the in-lane sort, in
// two halves: all NS loads are issued, but the lane only waits for the first NS / 2 of them (loads return in order), tests
// and sorts those -- a network of NS / 2 positions -- while the second half arrives, sorts that, and merges the two runs
// (second run taken backwards: a bitonic sequence, FusedBitonic<NS, 0>).  821 + 821 + 626 operations instead of the 2 184 of
// the 128-network, 4 % more -- for not standing idle through a whole gather: the network's first layer already touches every
// position, and the test for NaN samples in front of it needs every load anyway.  MEASURED SLOWER (round 4, sigma 512 x 4096^2:
// 10.98 against 10.13 ms, 14 instead of 4 spilled registers): with three waves per SIMD the gather of one wave already hides
// behind the networks of the other two.  Only A/B builds (NL_MLZ_HALVES) use it.
template <int LPP, int NS, int KPAD0, int NLOAD>
__device__ __forceinline__ int ml_gather_sort_halves(const float *frames, int64_t stride, int N, bool on, int64_t pix,
                                                     int role, float (&v)[NS])
{
    static_assert(NLOAD == NS && KPAD0 >= NS / 2 && KPAD0 <= NS, "a class that fills its lanes");
    constexpr int H = NS / 2;
    int nan_cnt = 0, spad = 0;
    int frame_bytes = (int)(stride * (int64_t)sizeof(float));
    asm volatile("" : "+s"(frame_bytes));                      // (see ml_gather_raw)
    const int voff = (int)((unsigned)(on ? pix : 0) * 4u) + role * frame_bytes;
    static_chunks<0, NS, 4>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        const int avail = min(max(N - k * LPP, 0), LPP);
        const char *gb = reinterpret_cast<const char *>(frames) + (int64_t)(k * LPP) * frame_bytes;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gb), 0, avail * frame_bytes, 0x00020000);
        v[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
    });
    auto nan_fix = [&](auto B, auto E) NL_INL {
        static_chunks<decltype(B)::value, decltype(E)::value, 8>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            nan_cnt = opaque(nan_cnt - ((0x7f800000 - (__float_as_int(v[k]) & 0x7fffffff)) >> 31));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[k]) : "v"(__builtin_inff()));   // NaN -> +Inf in place
        });
    };
    // ---- first half: every position holds a frame ----
    {
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        static_chunks<0, H / 4, 8>([&](auto K) NL_INL {
            constexpr int k = 4 * decltype(K)::value;
            t0 += v[k]; t1 += v[k + 1]; t2 += v[k + 2]; t3 += v[k + 3];
        });
        const float total = (t0 + t1) + (t2 + t3);
        if (__any(!(__builtin_fabsf(total) < __builtin_inff()))) nan_fix(std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
        float a[H];
        static_range<0, H>([&](auto K) NL_INL { a[decltype(K)::value] = v[decltype(K)::value]; });
        sort_network<H>(a);
        static_range<0, H>([&](auto K) NL_INL { v[decltype(K)::value] = a[decltype(K)::value]; });
    }
    // ---- second half: positions from KPAD0 on may lie beyond the last frame (-> +Inf, counted, out of the sum) ----
    {
        int lastp = opaque(N - 1) - role;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f, tp = 0.0f;
        static_range<KPAD0, NS>([&](auto K) NL_INL {
            constexpr int k = decltype(K)::value;
            const int pad = (lastp - k * LPP) >> 31;                       // all ones: no frame
            tp += __int_as_float(__float_as_int(v[k]) & ~pad);
            v[k] = __int_as_float((__float_as_int(v[k]) & ~pad) | (0x7f800000 & pad));
            spad -= pad;
        });
        static_chunks<H / 4, KPAD0 / 4, 8>([&](auto K) NL_INL {
            constexpr int k = 4 * decltype(K)::value;
            t0 += v[k]; t1 += v[k + 1]; t2 += v[k + 2]; t3 += v[k + 3];
        });
        static_range<KPAD0 / 4 * 4, KPAD0>([&](auto K) NL_INL { t0 += v[decltype(K)::value]; });
        const float total = ((t0 + tp) + t1) + (t2 + t3);
        if (__any(!(__builtin_fabsf(total) < __builtin_inff()))) nan_fix(std::integral_constant<int, H>{}, std::integral_constant<int, NS>{});
        float b[H];
        static_range<0, H>([&](auto K) NL_INL { b[decltype(K)::value] = v[H + decltype(K)::value]; });
        sort_network<H>(b);
        static_range<0, H>([&](auto K) NL_INL { v[H + decltype(K)::value] = b[H - 1 - decltype(K)::value]; });   // backwards
    }
    run_network<FusedBitonic<NS, 0>, NS>(v);
    return quad_sum<LPP>(NS - nan_cnt - spad);
}

template <int LPP, int NS, bool ENDS_ONLY, int KEEP = 16, int CH = 32, int NSL = NS, int KPAD0 = NS / 2, int NLOAD = NS, bool NT = false>
__device__ __forceinline__ int ml_gather_sorted(const float *frames, int64_t stride, int N, bool on, int64_t pix,
                                                int role, float (&v)[NS])
{
    const int n = ml_gather_raw<LPP, NS, KPAD0, NLOAD, NT>(frames, stride, N, on, pix, role, v);
    ml_sort_merge<LPP, NS, ENDS_ONLY, KEEP, CH, NSL>(v, role);
    return n;
}

}  // namespace nl
