// stack_fast_mlz_impl.hpp -- sigma / winsorized sigma clipping for 129..512 frames (2 or 4 lanes
// per pixel): the clipping rounds run on COLUMNS IN LDS instead of on the register column.
// Included by stack_fast_mlz_*.hip, which instantiate the kernel for their share of the frame-count
// classes: the kernel is compiled once per NTOP = frame count rounded up to a multiple of 16, so
// that the ranks it copies to LDS (the top of the column sits in the middle of a lane unless the
// stack fills its lanes) are compile-time register indices.
//
// stack_fast_ml.hip re-sums and re-counts its zones with static register indices, so every
// round costs a pass over all the positions any lane of the wave might need (and every lane of
// a pixel executes it).  Here the sort and the merge are the same, but afterwards each pixel
// writes to LDS, once,
//   * its KL lowest and KH highest ranks (the only samples a clip or a clamp can reach),
//   * suffix sums of (x-c) and (x-c)^2 over those columns, accumulated from the inner end
//     outwards at every 4th position (so an outlier never enters a sum it is not part of),
//   * the window of ranks the median can occupy,
// and keeps only the moments of everything in between (never clipped, never clamped).  A
// round is then a handful of per-lane LDS reads at data-dependent positions -- [slot][pixel]
// layout, conflict-free -- plus scalar work: the alive window [a, b) and the clamp positions
// are pointers that walk along the sorted columns, the sums come from the tables.  The cost of
// a round no longer depends on the frame count.
//
// Exactness is as in stack_fast.hip / DESIGN.md section 5: every decision is taken on a
// rigorous interval around the reference's fp32 value, undecidable pixels go to the exact
// replay, pixels whose clips or clamps leave the columns go to the generic pass.
// StackSigma: stack.go:372-436, StackWinsorSigma: stack.go:611-705.
#pragma once
#include <string>

#include "fast_ml_common.hpp"

namespace nl {

#ifdef NL_ROUND_STATS
__device__ unsigned long long nl_dbg_rounds_mlz[16];          // hand-over causes: [0] missing samples, [1] c2 >= 8, [2] d2 >= 8,
extern "C" int nl_debug_round_stats_mlz(unsigned long long *out, int reset)     // [3] low zone, [4] high zone, [5] shape -> exact
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nl_dbg_rounds_mlz), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nl_dbg_rounds_mlz), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define NL_STAT(i, x) atomicAdd(&nl_dbg_rounds_mlz[i], (unsigned long long)(x))
#else
#define NL_STAT(i, x) ((void)0)
#endif

namespace {

template <int LPP>
constexpr int mlz_block = LPP == 2 ? 128 : 256;     // threads per workgroup

// NTOP: ranks in use (the frame count rounded up to a multiple of 16; NT for a stack that fills its lanes)
template <int LPP, bool WINSOR, int NTOP>
struct MlzLayout {
    static constexpr int NS = kMlNS, NT = NS * LPP;
    static constexpr bool FULL = NTOP == NT;                    // the columns sit at the ends of lanes
    // a lane is dealt at most NTOP / LPP frames: its positions from there on hold +Inf, the in-lane sort is the
    // network of the next tabulated size (sort_tables.inc: 80, 96, 112, 128)
    static constexpr int NSL = NTOP / LPP <= 80 ? 80 : (NTOP / LPP <= 96 ? 96 : (NTOP / LPP <= 112 ? 112 : 128));
    // Winsorized kernels with four lanes per pixel: workgroups of 128 threads = 32 pixels, whose rounds phase is one
    // half-filled wave -- half the LDS per workgroup (30 KiB), so that a CU still holds a full complement of waves while
    // some of its workgroups are down to their rounds wave.  (With 64 pixels per workgroup -- 60 KiB, two workgroups per
    // CU -- the packed rounds measured 1 - 4 % SLOWER than unpacked ones; with 32: C3 tile kernel 3.98 -> 3.15 ms,
    // winsorized 300 / 400 frames 6.62 / 6.96 -> 5.12 / 5.50 ms.)
#ifdef NL_MLZ_SPACK
    static constexpr int BLOCK = (LPP == 4) ? 128 : mlz_block<LPP>;      // (A/B builds: the sigma kernels, too)
#else
    static constexpr int BLOCK = (WINSOR && LPP == 4) ? 128 : mlz_block<LPP>;
#endif
    static constexpr int PW = BLOCK / LPP;                      // pixels per workgroup = LDS row length (64 or 32)
    // alive window: a < ZLC, b > NTOP - ZHC; up to PADS missing samples (frames short of NTOP + NaNs)
    static constexpr int ZLC = 16, ZHC = FULL ? 24 : 32, PADS = FULL ? 15 : 23;
    static constexpr int CR = WINSOR ? 16 : 8;                  // ranks read per side and pass for the clip decisions
    // (winsorized, two lanes per pixel: the clamps sit at +-1.5 sigma, 6.7 % of the samples per side -- 17 +- 4 of 256, and
    // the low pointer must stay 8 ranks inside the column: with 32 ranks 6.5 % of the pixels of a 256-frame stack went to
    // the generic pass (550 k of 8.4 M: 3 ms of tail); the class that fills its lanes takes 40.  The other classes of
    // two lanes would need 41 KiB of LDS with 40 -- three instead of four workgroups per CU.)
    static constexpr int KL = WINSOR ? (LPP == 4 ? 64 : (FULL ? 40 : 32)) : 24;          // low column : ranks [0, KL)
    static constexpr int KH = (WINSOR ? (LPP == 4 ? 72 : 40) : 32) + (FULL ? 0 : 8);   // high column: ranks [NTOP-KH, NTOP)
    // SELECT: the columns and the median window are SELECTED from the lanes' sorted runs (select_ends / select_window below)
    // instead of read off a full cross-lane merge -- stacks that fill their lanes, plain sigma clipping
#ifdef NL_NO_SELECT
    static constexpr bool SELECT = false;                       // (A/B builds: tools/ab_lib.sh)
#else
    // (four lanes per pixel only: with two the full merge is one cross-lane stage and measures 4 % faster)
    static constexpr bool SELECT = FULL && !WINSOR && LPP == 4;
#endif
    // PACK: the clipping rounds run in ONE lane per pixel -- one wave for the workgroup's pixels, the others retire
    // (see the kernel).  A workgroup in its rounds still holds its LDS: the freed wave slots only fill if the CU has LDS
    // for more workgroups -- hence the smaller workgroups of the winsorized kernels (BLOCK above); the winsorized kernels
    // with two lanes per pixel (35 - 39 KiB for 64 pixels in two waves) stay unpacked.
#ifdef NL_MLZ_W2PACK
    static constexpr bool PACK = true;                           // (A/B builds: winsorized kernels with two lanes per pixel, too)
#else
    static constexpr bool PACK = !WINSOR || LPP == 4;
#endif
    static constexpr int KE = 32;                               // SELECT: ranks selected per end (>= KL, KH)
    static constexpr int KLS = SELECT ? KE : KL;                // LDS rows of the low column
    static constexpr int GL = KL / 4 + 1, GH = KH / 4 + 1;      // table entries
    // median window: kk = a + (b-a)/2 and kk-1 over all a < ZLC, b > NTOP - ZHC
    static constexpr int TOPW = ZHC / 2 + 2, BOTW = ZLC / 2 + 2, MW = TOPW + BOTW;
    static constexpr int W0 = NTOP / 2 - TOPW;                  // rank of window slot 0
    static constexpr int H0 = NTOP - KH;                        // rank of high-column slot 0
    // LDS rows, one float per pixel each
    // (SELECT stores whole selected runs with per-lane strides: 2 spare rows in front of the window, 8 behind it)
    // (SELECT: the low column's table sits in the rows of the selected ranks KL .. KE-1, which no round reads.
    // PS: 8 per-pixel scalars handed from the sorting phase to the rounds phase -- sample count, shift, fixed moments,
    // the squares taken off again, window flag, innermost ranks outside the columns; they share the rows of the tables,
    // which the rounds phase builds after it has read them.  32 KiB or less for the sigma kernels: five workgroups
    // per CU -- a workgroup in its rounds phase is one wave, but holds its LDS.)
#ifdef NL_MLZ_BIGLDS            // (A/B builds: tables and scalars in rows of their own, 35 KiB)
    static constexpr int XL = 0, XH = XL + KLS, SL1 = XH + KH, SL2 = SL1 + GL, SH1 = SL2 + GL, SH2 = SH1 + GH,
                         XW = SH2 + GH + (SELECT ? 2 : 0), PS = XW + MW + (SELECT ? 8 : 0), ROWS = PS + 8;
#else
    static constexpr int XL = 0, XH = XL + KLS, SL1 = SELECT ? XL + KL : XH + KH, SL2 = SELECT ? XH + KH : SL1 + GL,
                         SH1 = SL2 + GL, SH2 = SH1 + GH, XW = SH2 + GH + (SELECT ? 2 : 0), ROWS = XW + MW + (SELECT ? 8 : 0),
                         PS = SL2;
#endif
    static_assert(!SELECT || GL <= KLS - KL, "the low table fits behind the low column");
    static_assert(GL + GH >= 8, "rows for the per-pixel scalars");
    // roundings a term of the moment sums can see: fixed part (4 accumulators + quad adds), tables, assembly
    static constexpr int ROUNDINGS = (NS / 4 + 10 > KH + 4 ? NS / 4 + 10 : KH + 4) + 12;
    static_assert(NTOP % 16 == 0 && NTOP > NT / 2 && NTOP <= NT, "frame-count class");
    static_assert(KL % 8 == 0 && KH % 8 == 0 && KL >= ZLC + CR && KH >= ZHC + CR, "column sizes");
    static_assert(KL <= NS && KL <= H0 && W0 >= KL && W0 + MW <= H0 + KH, "columns and window inside the ranks in use");
    static_assert(!SELECT || (KL <= KE && KH == KE && TOPW <= 14 && BOTW <= 10 && PADS < KE), "selection sizes");
};

// The split pass of the SELECT class: which LDS rows travel from the sorting kernel to the rounds kernel (FastArgs::cols:
// one block of N rows x 64 pixels per workgroup, contiguous): the low column's KL ranks, the high column, the per-pixel scalars, the median window.
template <class L>
struct MlzSplit {
    static constexpr int N = L::KL + L::KH + 8 + L::MW;
    static constexpr int row(int g)
    {
        return g < L::KL ? L::XL + g : (g < L::KL + L::KH ? L::XH + (g - L::KL) : (g < L::KL + L::KH + 8 ? L::PS + (g - L::KL - L::KH) : L::XW + (g - L::KL - L::KH - 8)));
    }
    __device__ static __forceinline__ int row_rt(int g) { return row(g); }
};

// Where the rounds phase finds its rows.  MlzRows: the layout above, tables beside the columns (`tab` = `col`).
// MlzRowsPersistent (PHASE 3, the SELECT class): a compact buffer per block -- the KL / KH ranks of the columns the rounds
// read, the median window, the scalars and one TRASH row for what the selection stores beyond them (ranks KL .. KE-1 of the
// low end, the spare window slots) -- of which a workgroup holds TWO, and one table area shared by its rounds.
template <class L>
struct MlzRows {
    static constexpr int XL = L::XL, XH = L::XH, XW = L::XW, PS = L::PS, SL1 = L::SL1, SL2 = L::SL2, SH1 = L::SH1, SH2 = L::SH2;
    static constexpr int TRASH = -1, ROWS = L::ROWS, TROWS = 0;
};
template <class L>
struct MlzRowsPersistent {
    // (TRASH: the seventh scalar row -- only the winsorized kernels have a seventh scalar.  LDS is handed out in granules
    // of 1 280 bytes: with one more row per buffer a workgroup took 43 instead of 42 of them and a CU held two workgroups, not three)
    static constexpr int XL = 0, XH = XL + L::KL, XW = XH + L::KH, PS = XW + L::MW, TRASH = PS + 6, ROWS = PS + 8;
    static constexpr int SL1 = 0, SL2 = SL1 + L::GL, SH1 = SL2 + L::GL, SH2 = SH1 + L::GH, TROWS = SH2 + L::GH;
};

// ---- DPP minima / maxima: "mine" against the partner lane's "theirs" in ONE instruction ----
// (inline asm: the compiler's hazard recognizer does not look inside -- a VALU write of a register needs two
// wait states before a DPP read of it.  Every stage below starts with dpp_stage_begin() and only reads, through
// DPP, registers written before it.)
__device__ __forceinline__ void dpp_stage_begin()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
#define NL_DPP2(name, op, m0, m1, perm)                                                                              \
    __device__ __forceinline__ float name(float theirs, float mine)                                                  \
    {                                                                                                                \
        float r;                                                                                                     \
        asm(op " %0, " m0 "%1, " m1 "%2 quad_perm:" perm " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(theirs), "v"(mine)); \
        return r;                                                                                                    \
    }
NL_DPP2(min_swap1, "v_min_f32_dpp", "", "", "[1,0,3,2]")            // min(mine, partner ^ 1)
NL_DPP2(min_swap1_nn, "v_min_f32_dpp", "-", "-", "[1,0,3,2]")       // min(-mine, -partner ^ 1)
NL_DPP2(min_swap1_n, "v_min_f32_dpp", "-", "", "[1,0,3,2]")         // min(mine, -partner ^ 1)
NL_DPP2(min_swap2, "v_min_f32_dpp", "", "", "[2,3,0,1]")            // min(mine, partner ^ 2)
NL_DPP2(max_swap2, "v_max_f32_dpp", "", "", "[2,3,0,1]")            // max(mine, partner ^ 2)
NL_DPP2(min_mirror_n, "v_min_f32_dpp", "-", "", "[3,2,1,0]")        // min(mine, -partner 3 - lane)
#undef NL_DPP2

// ascending half-cleaner cascade of a bitonic sequence (distances D, D/2 ... 1), raw min / max (the selection
// front end runs the fused tables FusedBitonic<64 / 32 / 16, 0> of sort_tables.inc instead: 30 % fewer operations)
template <int N, int D>
__device__ __forceinline__ void clean_raw(float (&x)[N])
{
    if constexpr (D >= 1) {
        static_range<0, N / 2>([&](auto T) NL_INL {
            constexpr int t = decltype(T)::value;
            constexpr int i = ((t & ~(D - 1)) << 1) | (t & (D - 1));
            constexpr int l = i | D;
            const float lo = min_raw(x[i], x[l]), hi = max_raw(x[i], x[l]);
            x[i] = lo;
            x[l] = hi;
        });
        clean_raw<N, (D >> 1)>(x);
    }
}

// SELECT front end (stacks that fill their lanes, plain sigma clipping).  After the in-lane sort every lane
// holds a sorted run of NS samples; the clipping rounds only ever look at
//   * the KL lowest / KH highest ranks of the pixel: they lie among the KE lowest / highest samples of the
//     lanes' runs -- ALWAYS -- so a bitonic selection over those (KE per lane, two cross-lane stages) yields
//     them; even lanes select the low end while the odd lanes select the high end on NEGATED samples (one
//     instruction stream: min everywhere),
//   * the ranks the median can take: the middle 64 samples of every run are merged (lower halves in the plain
//     lanes, upper halves, negated, in the others) and only the 16 ranks either side of the middle are sorted
//     out.  Rank = position in that merge + LPP * 32 holds iff no sample below a run's middle 64 exceeds, and
//     none above them is below, the window: checked, pixels that fail go to the generic pass (a run's 25 % /
//     75 % quantile would have to cross the pixel's median: 6 sigma of its sampling noise at 128 samples),
//   * the moments of everything between the columns: summed with the ends of the runs clamped to the
//     innermost column values [t_lo, t_hi] -- a sample inside a column contributes a known constant, taken off
//     afterwards (its magnitude is the bulk's, not an outlier's: no cancellation to speak of; the extra
//     rounding is covered by q_cancel in the error bound).
// Cost: about 1 700 instructions instead of 2 500 for merge + staging + masked moments.
template <class L, int LPP, int NS, class V = MlzRows<L>>
__device__ __forceinline__ void select_ends(const float (&v)[NS], int role, float *col, float &t_lo, float &t_hi)
{
    constexpr int PW = L::PW, KE = L::KE;
    const bool odd = (role & 1) != 0;
    const int sgn = odd ? (int)0x80000000 : 0;
    float z[KE];
    dpp_stage_begin();
    static_range<0, KE>([&](auto I) NL_INL {
        constexpr int i = decltype(I)::value;
        const float lo = min_swap1(v[KE - 1 - i], v[i]);                        // KE smallest of the pair's low ends
        const float hn = min_swap1_nn(v[NS - KE + i], v[NS - 1 - i]);           // -(KE largest of the pair's high ends)
        z[i] = odd ? hn : lo;
    });
    run_network<FusedBitonic<KE, 0>, KE>(z);
    if constexpr (LPP == 4) {
        dpp_stage_begin();
        static_range<0, KE / 2>([&](auto I) NL_INL {                            // (pairs, in place: short live ranges)
            constexpr int i = decltype(I)::value, j = KE - 1 - i;
            const float a = min_swap2(z[j], z[i]), b = min_swap2(z[i], z[j]);
            z[i] = a;
            z[j] = b;
        });
        run_network<FusedBitonic<KE, 0>, KE>(z);
    }
    // even lanes: z[i] = rank i; odd lanes: z[i] = -(rank NTOP-1-i).  Rows: low column XL + i, high column XH + KE-1-i
    float *dst = col + (odd ? (V::XH + KE - 1) * PW : V::XL * PW);
    const int step = odd ? -PW : PW;
    static_range<0, KE>([&](auto I) NL_INL {
        constexpr int i = decltype(I)::value;
        float *at = dst + i * step;
        if constexpr (V::TRASH >= 0 && i >= L::KL) at = odd ? at : col + V::TRASH * PW;      // (no row for the low ranks KL .. KE-1)
        *at = __int_as_float(__float_as_int(z[i]) ^ sgn);
    });
    // innermost column values, in every lane of the pixel
    const float tl = z[L::KL - 1];                                              // rank KL-1 (even lanes)
    const float th = __int_as_float(__float_as_int(z[KE - 1]) ^ (int)0x80000000);          // rank NTOP-KE (odd lanes)
    t_lo = __int_as_float(quad_bcast<LPP, 0>(__float_as_int(tl)));
    t_hi = __int_as_float(quad_bcast<LPP, 1>(__float_as_int(th)));
}

// the median window; consumes the runs' middles (v is dead afterwards)
template <class L, int LPP, int NS, class V = MlzRows<L>>
__device__ __forceinline__ void select_window(float (&v)[NS], int role, float *col, bool &window_ok)
{
    constexpr int PW = L::PW;
    const bool odd = (role & 1) != 0;
    const int sgn = odd ? (int)0x80000000 : 0;
    constexpr int C0 = NS / 4, CN = NS / 2;                                     // candidates: run positions [32, 96)
    // largest sample below / smallest above the candidates of any run of the pixel
    float below = v[C0 - 1], above = v[C0 + CN];
    below = fmaxf(below, dpp_f<kSwap1>(below));
    above = fminf(above, dpp_f<kSwap1>(above));
    if constexpr (LPP == 4) {
        below = fmaxf(below, dpp_f<kSwap2>(below));
        above = fminf(above, dpp_f<kSwap2>(above));
    }
    float t[CN];
    static_range<0, CN>([&](auto I) NL_INL {
        constexpr int i = decltype(I)::value;
        t[i] = __int_as_float(__float_as_int(v[C0 + i]) ^ sgn);                 // odd lanes work on negated samples
    });
    dpp_stage_begin();
    static_range<0, CN / 2>([&](auto I) NL_INL {
        constexpr int i = decltype(I)::value, j = CN - 1 - i;
        // plain lanes: lower half of the pair's 128 candidates, the others: -(upper half)
        const float a = min_swap1_n(t[j], t[i]), b = min_swap1_n(t[i], t[j]);
        t[i] = a;
        t[j] = b;
    });
    float w[16];
    float kept_min = 0.0f;                                  // LPP == 4: smallest of a lane's kept candidates (its own sign)
    if constexpr (LPP == 4) {
        // Round 3 sorted a lane's 64 candidates (the bitonic lower half of its pair's 128) completely.  Only their upper
        // 32 can reach the 16 ranks below the middle of the pixel that this selection is after -- unless a pair of runs
        // holds fewer than 96 of the 242 samples below the window, 4.5 sigma of its sampling noise -- so the half-cleaner
        // keeps those (one max per element, a bitonic 32-sequence), the cascade sorts 32 instead of 64 values, and the
        // stages behind it run over half the positions: 220 instructions less per wave.  Every dropped candidate is at
        // most the smallest one kept, so "smallest kept <= lowest window value" proves that none of them belongs into
        // the window (checked below with the runs' ends; a pixel that fails goes to the generic pass as before).
        constexpr int CK = CN / 2;
        float keep[CK];
        static_range<0, CK>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            keep[i] = max_raw(t[i], t[i + CK]);
        });
        run_network<FusedBitonic<CK, 0>, CK>(keep);        // sorted: lanes 0 / 2 the upper 32 of their lower halves, 1 / 3 the same of -(upper halves)
        kept_min = keep[0];
        dpp_stage_begin();
        static_range<0, CK>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            keep[i] = min_mirror_n(keep[i], keep[i]);      // lanes 0, 2: positions 32 .. 95 of the (bitonic) lower 128 of the 256 candidates; 1, 3: of -(upper 128)
        });
        dpp_stage_begin();
        // the 32 largest of those 64 (a contiguous stretch of a bitonic sequence is bitonic): partner lane ^ 2, mirrored
        // index; of those (bitonic again) the 16 largest
        float m2[CK];
        static_range<0, CK>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            m2[i] = max_swap2(keep[CK - 1 - i], keep[i]);
        });
        static_range<0, 16>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            w[i] = max_raw(m2[i], m2[i + 16]);
        });
    } else {
        // two lanes: t is the (bitonic) lower / -(upper) 64 of the 128 candidates: its 16 largest
        float m2[CN / 2];
        static_range<0, CN / 2>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            m2[i] = max_raw(t[i], t[i + CN / 2]);
        });
        static_range<0, 16>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            w[i] = max_raw(m2[i], m2[i + 16]);
        });
    }
    run_network<FusedBitonic<16, 0>, 16>(w);
    // plain lanes: w[j] = candidate rank (LPP * 32 - 16) + j, i.e. window slot j - 2 for j >= 2; the other lanes:
    // w[j] = -(candidate rank LPP * 32 + 15 - j), window slot 14 + 15 - j (slots up to MW - 1: j >= 6).  The rows
    // outside the window are spare (MlzLayout).
    float *dst = col + (odd ? (V::XW + 29) * PW : (V::XW - 2) * PW);
    const int step = odd ? -PW : PW;
    static_range<0, 16>([&](auto J) NL_INL {
        constexpr int j = decltype(J)::value;
        float *at = dst + j * step;
        if constexpr (V::TRASH >= 0 && j < 2) at = col + V::TRASH * PW;                      // (slots -2, -1 / 29, 28: spare)
        else if constexpr (V::TRASH >= 0 && j < 6) at = odd ? col + V::TRASH * PW : at;      // (slots 27 .. 24: spare)
        *at = __int_as_float(__float_as_int(w[j]) ^ sgn);
    });
    // window slot 0 must not be below `below`, slot MW-1 not above `above`
    const float w_first = __int_as_float(quad_bcast<LPP, 0>(__float_as_int(w[2])));
    const float w_last = __int_as_float(quad_bcast<LPP, 1>(__float_as_int(w[6]) ^ (int)0x80000000));
    window_ok = below <= w_first && w_last <= above;
    if constexpr (LPP == 4) {
        // ... and no dropped candidate may belong into the window: the smallest kept one of either lane of this parity
        // (negated samples in the odd lanes) against the window's end on this side
        const float km = fmaxf(kept_min, dpp_f<kSwap2>(kept_min));
        const bool ok = km <= (odd ? -w_last : w_first);
        window_ok = window_ok && quad_or<LPP>(ok ? 0 : 1) == 0;
    }
}

// LDS operations of one wave complete in order; the clobber keeps the compiler from moving
// loads of other lanes' stores across this point
__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace

#ifndef NL_MLZ_STAGGER
#define NL_MLZ_STAGGER 3        // x 8 128 cycles per third of the grid
#endif
#ifndef NL_MLZ_WINSOR_WAVES
#define NL_MLZ_WINSOR_WAVES 2
#endif
// PHASE (the SELECT class only, see MlzSplit below): 0 = the whole pass in one kernel; 1 = the sorting phase, whose
// columns go to FastArgs::cols instead of staying in LDS; 2 = the rounds phase over those columns, one wave per workgroup.
// PHASE 3: a wave waits for a flag of its workgroup.  (In practice the flag is long set; a wait that never ends -- a bug --
// traps instead of hanging the device.)
__device__ __forceinline__ void mlz_spin_until(unsigned *flag, unsigned target)
{
    for (int i = 0; i < (1 << 22); i++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return;
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_trap();
}

template <int LPP, bool WINSOR, int NTOP, int PHASE = 0>
__global__ __launch_bounds__(PHASE == 2 ? 64 : mlz_block<LPP>)
__attribute__((amdgpu_waves_per_eu(PHASE == 2 ? 1 : (WINSOR ? NL_MLZ_WINSOR_WAVES : 3), 8)))
void stack_sigma_mlz_kernel(StackArgs p, FastArgs q)
{
    using L = MlzLayout<LPP, WINSOR, NTOP>;
    using SP = MlzSplit<L>;
    using V = std::conditional_t<PHASE == 3, MlzRowsPersistent<L>, MlzRows<L>>;
    static_assert(PHASE == 0 || (L::SELECT && L::PACK && L::PW == 64), "the split and the persistent pass exist for the SELECT class");
    static_assert(PHASE != 3 || (!WINSOR && (2 * V::ROWS + V::TROWS) * L::PW * 4 + 64 <= 42 * 1280), "persistent pass: three workgroups per CU");
    constexpr int NS = L::NS, PW = L::PW, KL = L::KL, KH = L::KH, CR = L::CR, H0 = L::H0, W0 = L::W0;
    // (PHASE 3: two column buffers and one table area)
    __shared__ float lds[(PHASE == 3 ? 2 * V::ROWS + V::TROWS : L::ROWS) * PW];
    __shared__ unsigned s_done[2], s_freed[2];             // PHASE 3: waves that wrote / rounds that finished, per buffer
    __shared__ unsigned s_blk[4], s_seq;                   // PHASE 3: the workgroup's block of trip k (slot k & 3), trips published
    __shared__ int s_lo[4], s_hi[4];                       // (kernels whose rounds run in every wave: their clip counts)
    if constexpr (PHASE != 2) fused_prologue_dominant(p);

    // Two phases.  SORTING: LPP lanes per pixel, every wave of the workgroup -- gather, in-lane sort, merge / selection,
    // columns + median window + the moments between the columns to LDS.  ROUNDS: ONE lane per pixel, i.e. one wave for
    // the workgroup's 64 pixels (the others retire at the barrier) -- tables, clipping / winsorization rounds, results.
    // Round 3 ran the rounds in all LPP lanes of a pixel (same values, LPP times the instructions: a round is per-lane
    // LDS reads at data-dependent rows plus scalar-like arithmetic, nothing the lanes of a pixel could share out);
    // a fifth (sigma) to a half (winsorized) of the kernel's instructions were those.  The wave that stays is picked
    // by the workgroup index so that a CU's SIMDs share the rounds evenly.
    const int lane = threadIdx.x & 63;
    // ---- sorting phase of block `blk` (PW pixels); its rows go to `colbase`; `before_store` runs in front of the first store ----
    auto sorting_phase = [&](const int64_t blk, float *const colbase, auto &&before_store) NL_INL {
    // (PHASE 3 runs this in a loop: everything derived from the thread index would be hoisted out of it -- some thirty
    // registers, i.e. spills to scratch and their reloads on the waves' critical path (measured: 17.9 instead of 10.1 ms) --
    // so the index goes through an opaque register per trip and the few shifts and masks are redone)
    int tid = (int)threadIdx.x;
    if constexpr (PHASE == 3) asm volatile("" : "+v"(tid));
    const int role = tid % LPP;
    float *col = colbase + tid / LPP;                      // element r of this pixel: col[r * PW]
    const int64_t pix = blk * PW + tid / LPP;
    // (PHASE 3: the wait for the buffer -- over long before -- stands HERE, not in front of the first store: a loop in the
    // middle of the sorting phase cuts its one basic block in two and cost 14 more spilled registers)
    if constexpr (L::SELECT) before_store();
    const bool on = pix < p.npix;
    int N = p.n_frames;
    // (PHASE 3 calls this in a loop: re-read through an opaque register, or the per-frame scalar selects that depend on the
    // frame count are hoisted out of the loop and spilled)
    if constexpr (PHASE == 3) asm volatile("" : "+s"(N));
    float v[NS];
    // a stack that fills its lanes: the last merge orders the 32 lowest / highest ranks of every lane
    // (columns of the plain sigma kernel, median window); the winsorized columns are longer, and the
    // columns of a shorter stack sit inside the lanes: full merge
    int n;
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
    const unsigned long long sp0 = __builtin_readcyclecounter();
    unsigned long long sp1 = sp0, sp2 = sp0, sp3 = sp0;
#endif
    if constexpr (L::SELECT) {
        // (the lanes' runs are not merged: select_ends / select_window)
#ifdef NL_MLZ_HALVES             // (A/B builds: sort the first 64 positions while the other 64 loads are in flight -- 8 % slower)
        n = ml_gather_sort_halves<LPP, NS, (NTOP - 16) / LPP, NTOP / LPP>(p.frames, p.stride, N, on, pix, role, v);
#else
        n = ml_gather_raw<LPP, NS, (NTOP - 16) / LPP, NTOP / LPP, LPP == 2>(p.frames, p.stride, N, on, pix, role, v);      // (nt at two lanes per pixel: fast_ml_common.hpp)
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
        __builtin_amdgcn_sched_barrier(0); sp1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        sort_network<NS>(v);
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
        __builtin_amdgcn_sched_barrier(0); sp2 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
#endif
    } else {
        n = ml_gather_sorted<LPP, NS, L::FULL && !WINSOR, 32, 32, L::NSL, (NTOP - 16) / LPP, NTOP / LPP, LPP == 2 && !WINSOR>(p.frames, p.stride, N, on, pix,
                                                                                                              role, v);
    }

    // ---- columns and median window to LDS ----
    // (no divergent branches while the column is in registers -- they cost the compiler's register
    // allocation over a hundred spills: the owning lane's value is broadcast inside the pixel's quad
    // and every lane of the pixel stores it, same value to the same address)
    auto bcast_f = [](auto R, float x) NL_INL {
        return __int_as_float(quad_bcast<LPP, decltype(R)::value>(__float_as_int(x)));
    };
    bool window_ok = true;
    float d_fix = 0.0f, q_fix = 0.0f;                      // moments of the ranks between the columns
    float q_cancel = 0.0f;                                 // SELECT: squares taken off again (enters the rounding bound)
    float c_sel = 0.0f;
    if constexpr (L::SELECT) {
        // shift: the middle of the first lane's run (any value near the bulk works)
        c_sel = bcast_f(std::integral_constant<int, 0>{}, v[NS / 2]);
        float t_lo, t_hi;                                  // innermost values of the low / high column
        select_ends<L, LPP, NS, V>(v, role, col, t_lo, t_hi);
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
        __builtin_amdgcn_sched_barrier(0); sp3 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
        // moments of every sample of the pixel, the ends of the runs clamped to [t_lo, t_hi]: the KL + KH samples
        // of the columns count as t_lo / t_hi (a missing sample, +Inf, as t_hi) and are taken off again
        {
            const float c = c_sel;
            float d0 = 0, d1 = 0, d2 = 0, d3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            static_chunks<0, NS / 4, 4>([&](auto K4) NL_INL {
                constexpr int k = 4 * decltype(K4)::value;
                auto term = [&](auto KK) NL_INL {
                    constexpr int kk = decltype(KK)::value;
                    if constexpr (kk < L::KE) return max_raw(v[kk], t_lo) - c;
                    else if constexpr (kk >= NS - L::KE) return min_raw(v[kk], t_hi) - c;
                    else return v[kk] - c;
                };
                const float e0 = term(std::integral_constant<int, k>{}), e1 = term(std::integral_constant<int, k + 1>{});
                const float e2 = term(std::integral_constant<int, k + 2>{}), e3 = term(std::integral_constant<int, k + 3>{});
                d0 += e0; d1 += e1; d2 += e2; d3 += e3;
                q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
                q2 = __builtin_fmaf(e2, e2, q2); q3 = __builtin_fmaf(e3, e3, q3);
            });
            const float d_all = quad_sum<LPP>((d0 + d1) + (d2 + d3));
            const float q_all = quad_sum<LPP>((q0 + q1) + (q2 + q3));
            const float e_lo = t_lo - c, e_hi = t_hi - c;
            q_cancel = (float)KL * (e_lo * e_lo) + (float)KH * (e_hi * e_hi);
            d_fix = d_all - ((float)KL * e_lo + (float)KH * e_hi);
            q_fix = q_all - q_cancel;
        }
        select_window<L, LPP, NS, V>(v, role, col, window_ok);
    } else {
    before_store();
    static_range<0, KL>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value;
        col[(V::XL + k) * PW] = bcast_f(std::integral_constant<int, 0>{}, v[k]);
    });
    static_range<0, KH>([&](auto K) NL_INL {                          // rank H0 + k: lane rank / NS, register rank % NS
        constexpr int k = decltype(K)::value, r = H0 + k;
        col[(V::XH + k) * PW] = bcast_f(std::integral_constant<int, r / NS>{}, v[r % NS]);
    });
    static_range<0, L::MW>([&](auto K) NL_INL {
        constexpr int k = decltype(K)::value, r = W0 + k;
        col[(V::XW + k) * PW] = bcast_f(std::integral_constant<int, r / NS>{}, v[r % NS]);
    });
    }
    lds_settle();

    // shift c = first-pass median (any value near the bulk works)
    float c = col[(V::XW + min(max((n >> 1) - W0, 0), L::MW - 1)) * PW];
    if constexpr (L::SELECT) c = c_sel;

    // ---- moments of the ranks between the columns (never clipped, never clamped) ----
    // blocks of 8 registers; a block of lane `role` counts if its ranks lie in [KL, H0) (both multiples of 8):
    // the rest is a column, or padding above the ranks in use
    if constexpr (!L::SELECT) {
        float da[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qa[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int off = role * NS - KL;
        static_chunks<0, NS / 8, 2>([&](auto J) NL_INL {
            constexpr int j = decltype(J)::value;
            float d0 = 0.0f, d1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
            static_range<0, 4>([&](auto U) NL_INL {
                constexpr int k = 8 * j + 2 * decltype(U)::value;
                const float e0 = v[k] - c, e1 = v[k + 1] - c;
                d0 += e0; d1 += e1;
                q0 = __builtin_fmaf(e0, e0, q0); q1 = __builtin_fmaf(e1, e1, q1);
            });
            const bool inc = (unsigned)(8 * j + off) < (unsigned)(H0 - KL);
            da[j & 3] += inc ? d0 + d1 : 0.0f;
            qa[j & 3] += inc ? q0 + q1 : 0.0f;
        });
        d_fix = quad_sum<LPP>((da[0] + da[1]) + (da[2] + da[3]));
        q_fix = quad_sum<LPP>((qa[0] + qa[1]) + (qa[2] + qa[3]));
    }
    // rank KL (first above the low column) and rank H0-1 (last below the high column)
    float x_in_lo = 0.0f, x_in_hi = 0.0f;                 // (only the winsorized rounds look at them)
    if constexpr (WINSOR) {
        x_in_lo = bcast_f(std::integral_constant<int, KL / NS>{}, v[KL % NS]);
        x_in_hi = bcast_f(std::integral_constant<int, (H0 - 1) / NS>{}, v[(H0 - 1) % NS]);
    }
    // the pixel's scalars for the rounds phase (every lane of the pixel stores the same values)
    col[(V::PS + 0) * PW] = __int_as_float(n);
    col[(V::PS + 1) * PW] = c;
    col[(V::PS + 2) * PW] = d_fix;
    col[(V::PS + 3) * PW] = q_fix;
    col[(V::PS + 4) * PW] = q_cancel;
    col[(V::PS + 5) * PW] = __int_as_float(window_ok ? 1 : 0);
    if constexpr (WINSOR) {
        col[(V::PS + 6) * PW] = x_in_lo;
        col[(V::PS + 7) * PW] = x_in_hi;
    }
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
    if (threadIdx.x == 0) {
        const unsigned long long sp4 = __builtin_readcyclecounter();
        NL_STAT(8, sp1 - sp0); NL_STAT(9, sp2 - sp1); NL_STAT(10, sp3 - sp2); NL_STAT(11, sp4 - sp3); NL_STAT(12, 1);
    }
#endif
    };   // ---- end of the sorting phase ----

    // ---- rounds phase of block `blk`: columns at `colbase`, tables at `tabbase` ----
    auto rounds_phase = [&](const int64_t blk, float *const colbase, float *const tabbase) NL_INL {
    int c_lo_total = 0, c_hi_total = 0;
    int tid = (int)threadIdx.x;
    if constexpr (PHASE == 3) asm volatile("" : "+v"(tid));        // (see the sorting phase)
    const int lane = tid & 63;
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
    const unsigned long long exp_t0 = __builtin_readcyclecounter();      // (timing experiment: cycles of the rounds phase)
#endif
    // ---- rounds phase: PACK: lane = pixel of the workgroup; else every lane of a pixel runs its rounds ----
    const int role = L::PACK ? 0 : (int)(threadIdx.x % LPP);         // (role 0 reports)
    const int slot_px = L::PACK ? min(lane, PW - 1) : (int)(threadIdx.x / LPP);      // (PW < 64: the upper lanes idle)
    float *col = colbase + slot_px;
    float *tab = tabbase + slot_px;                        // (the tables: beside the columns, or the workgroup's table area)
    const int64_t pix = blk * PW + slot_px;
    const bool on = pix < p.npix && (!L::PACK || lane < PW);
    const int n = __float_as_int(col[(V::PS + 0) * PW]);
    const float c = col[(V::PS + 1) * PW];
    const float d_fix = col[(V::PS + 2) * PW], q_fix = col[(V::PS + 3) * PW];
    const float q_cancel = col[(V::PS + 4) * PW];
    const bool window_ok = __float_as_int(col[(V::PS + 5) * PW]) != 0;
    float x_in_lo = 0.0f, x_in_hi = 0.0f;
    if constexpr (WINSOR) {
        x_in_lo = col[(V::PS + 6) * PW];
        x_in_hi = col[(V::PS + 7) * PW];
    }

    bool active = on && n > 0;
    // the alive window must keep its ends inside the columns: at most PADS missing samples (SELECT: and the
    // median window must have come out of the runs' middles)
    bool to_generic = active && (!(n > NTOP - 1 - L::PADS) || !window_ok);
    if (to_generic && role == 0) NL_STAT(0, 1);
    active = active && !to_generic;
    bool to_exact = false;

    // ---- tables: sums from the inner end of each column outwards, every 4th position ----
    // (the whole column is read before the first table entry is stored: the compiler cannot move a read of `col` across a
    // store to it, and four reads per LDS round trip made the tables a third of this phase's latency)
    {
        float t[KL], th[KH];
        static_range<0, KL>([&](auto K) NL_INL { t[decltype(K)::value] = col[(V::XL + decltype(K)::value) * PW]; });
        static_range<0, KH>([&](auto K) NL_INL { th[decltype(K)::value] = col[(V::XH + decltype(K)::value) * PW]; });
        {
        float s1 = 0.0f, s2 = 0.0f;
        tab[(V::SL1 + KL / 4) * PW] = 0.0f;
        tab[(V::SL2 + KL / 4) * PW] = 0.0f;
        static_range<0, KL / 4>([&](auto G) NL_INL {
            constexpr int g = KL / 4 - 1 - decltype(G)::value;
            static_range<0, 4>([&](auto U) NL_INL {
                const float e = t[4 * g + 3 - decltype(U)::value] - c;
                s1 += e;
                s2 = __builtin_fmaf(e, e, s2);
            });
            tab[(V::SL1 + g) * PW] = s1;                   // sum over k >= 4g
            tab[(V::SL2 + g) * PW] = s2;
        });
        }
        float s1 = 0.0f, s2 = 0.0f;
        tab[(V::SH1 + 0) * PW] = 0.0f;
        tab[(V::SH2 + 0) * PW] = 0.0f;
        static_range<0, KH / 4>([&](auto G) NL_INL {
            constexpr int g = decltype(G)::value;
            static_range<0, 4>([&](auto U) NL_INL {
                const float e = th[4 * g + decltype(U)::value] - c;
                s1 += e;
                s2 = __builtin_fmaf(e, e, s2);
            });
            // sum over local t < 4(g+1); missing samples (+Inf) only reach entries that are never read (4g' <= b)
            tab[(V::SH1 + g + 1) * PW] = s1;
            tab[(V::SH2 + g + 1) * PW] = s2;
        });
    }
    lds_settle();

    float res = p.ref_loc;
    int c_lo = 0, c_hi = 0;
    int rnd = 0;                                           // clipping rounds decided so far (StackArgs::bounds)
    int a = 0, b = n;                                      // survivors = sorted ranks [a, b)
    // (+8: the means are taken with a hardware reciprocal of the sample count, <= 1 ulp, instead of IEEE divisions --
    // a round is a chain of dependent operations in ONE wave that holds the workgroup's LDS and wave slots, and the two
    // divisions and two correctly rounded square roots were a fifth of its instructions)
#ifdef NL_MLZ_IEEE_ROUNDS
    constexpr float kErrF = (float)(2 * L::ROUNDINGS + 8);
#else
    constexpr float kErrF = (float)(2 * L::ROUNDINGS + 16);
#endif

    while (__any(active)) {
        const int cnt = b - a;
        const float fcnt = (float)cnt;
        const int al = min(max(a, 0), KL - CR);                      // (clamped for the address only)
        const int bl = min(max(b - H0, CR), KH);              // local end of the alive part of the high column
        // ---- reads: CR ranks from each end of the alive window (xl[i] = rank a+i, xh[i] = rank b-1-i),
        // the tables, the median ----
        float xl[CR], xh[CR];
        {
            const float *pl = col + (V::XL + al) * PW;
            const float *ph = col + (V::XH + bl - 1) * PW;
            static_range<0, CR>([&](auto I) NL_INL {
                xl[decltype(I)::value] = pl[decltype(I)::value * PW];
                xh[decltype(I)::value] = *(ph - decltype(I)::value * PW);
            });
        }
        const int ga = al >> 2, gb = bl >> 2;
        const float sl1 = tab[(V::SL1 + ga + 1) * PW], sl2 = tab[(V::SL2 + ga + 1) * PW];
        const float sh1 = tab[(V::SH1 + gb) * PW], sh2 = tab[(V::SH2 + gb) * PW];
        const int kk = a + (cnt >> 1);
        const int wi0 = min(max(kk - W0, 1), L::MW - 1);
        const float upper = col[(V::XW + wi0) * PW], lower = col[(V::XW + wi0 - 1) * PW];
        const float median = (cnt & 1) ? upper : 0.5f * (lower + upper);       // qsort.go:68-82

        // ---- moments of the survivors: fixed part + tables + the partial groups at a and b ----
        float dz = 0.0f, qz = 0.0f;
        {
            const int nlp = 4 - (al & 3);                  // ranks al .. 4(ga+1)-1
            static_range<0, 4>([&](auto I) NL_INL {
                constexpr int i = decltype(I)::value;
                const float e = (i < nlp) ? xl[i] - c : 0.0f;
                dz += e;
                qz = __builtin_fmaf(e, e, qz);
            });
            const int nhp = bl & 3;                        // local 4 gb .. bl-1
            static_range<0, 3>([&](auto I) NL_INL {
                constexpr int i = decltype(I)::value;
                const float e = (i < nhp) ? xh[i] - c : 0.0f;
                dz += e;
                qz = __builtin_fmaf(e, e, qz);
            });
        }
        const float dsum = (d_fix + dz) + (sl1 + sh1);
        const float qsum = (q_fix + qz) + (sl2 + sh2);
#ifdef NL_MLZ_IEEE_ROUNDS
        const float inv_cnt = 1.0f / fcnt;
        const float delta = dsum / fcnt;                   // mean~ - c
#else
        const float inv_cnt = __builtin_amdgcn_rcpf(fcnt);
        const float delta = dsum * inv_cnt;                // mean~ - c
#endif
        const float m = c + delta;
#ifdef NL_MLZ_IEEE_ROUNDS
        const float aa = qsum / fcnt;                      // E[(x-c)^2]~
#else
        const float aa = qsum * inv_cnt;                   // E[(x-c)^2]~
#endif
        const float bb = delta * delta;
        const float var = fmaxf(aa - bb, 0.0f);

        // ---- bracket the reference's stddev ----
        const float amax = fmaxf(fabsf(xl[0]), fabsf(xh[0]));
        // (SELECT: d_fix / q_fix carry the rounding of the column samples that were summed at [t_lo, t_hi] and
        // taken off again: their squares join the magnitude the bound scales with)
        // (factor: the mean's rounding scales with the mean of |e| over the alive samples AND the KL + KH column samples
        // that were taken off again -- by Cauchy-Schwarz its square is at most (1 + 56 / cnt) (aa + q_cancel / cnt),
        // cnt >= 457, and 2 |delta| x <= delta^2 + x^2: (1 + 1.1225) / 2 = 1.062 of the bound without the columns)
        const float err_o = L::SELECT ? (1.07f * kErrF) * kU * ((aa + bb) + q_cancel * inv_cnt) : kErrF * kU * (aa + bb);
        const float eps_r = 1.02f * (fcnt + 8.0f) * kU;
        const float e_m = 1.02f * (fcnt + 2.0f) * kU * amax;
        const float v_up = var + err_o;
        const float v_dn = fmaxf(var - err_o, 0.0f);
        const float v_hi = v_up + v_up * eps_r + e_m * e_m;
        const float v_lo = fmaxf(v_dn - v_dn * eps_r, 0.0f);
#ifdef NL_MLZ_IEEE_ROUNDS
        float s_max = __fsqrt_rn(v_hi) * (1.0f + 4.0f * kU);
        float s_min = __fsqrt_rn(v_lo) * (1.0f - 4.0f * kU);
#else
        // hardware square root: <= 1 ulp = 2u, flushes denormals (the absolute term; a flushed lower end is 0, still a lower end)
        float s_max = __builtin_amdgcn_sqrtf(v_hi) * (1.0f + 6.0f * kU) + 4.0e-19f;
        float s_min = __builtin_amdgcn_sqrtf(v_lo) * (1.0f - 6.0f * kU);
#endif
        bool bail = !(v_hi < 3.0e38f);

        if constexpr (WINSOR) {
            // ---- winsorized stddev (stack.go:646-672) as an interval, WinsorInterval in fast_common.hpp ----
            // jl = first rank of the low column that is not below the (tightest) low clamp Lp,
            // jh = local end of the ranks of the high column that are not above the high clamp Hm:
            // both clamps only tighten inside one loop, so the pointers only move inwards
            WinsorInterval wi;
            wi.start(s_min, s_max);
            bool inner = active && !bail;
            int jl = al, jh = bl;
            bool first = true;
            while (__any(inner)) {
                wi.next_clamp(median, xl[0], xh[0]);
                if (first) {
                    // coarse start: every 8th rank of the columns
                    int t_lo = 0, t_hi = 0;
                    static_range<0, KL / 8>([&](auto G) NL_INL {
                        t_lo += (col[(V::XL + 8 * decltype(G)::value + 7) * PW] < wi.Lp) ? 1 : 0;
                    });
                    static_range<0, KH / 8>([&](auto G) NL_INL {
                        t_hi += (col[(V::XH + 8 * decltype(G)::value) * PW] > wi.Hm) ? 1 : 0;
                    });
                    jl = max(jl, 8 * t_lo);
                    jh = min(jh, KH - 8 * t_hi);
                    first = false;
                }
                // walk: 8 candidates per step and side
                bool more = inner;
                while (__any(more)) {
                    const int jlc = min(jl, KL - 8), jhc = max(jh, 8);
                    const float *pl = col + (V::XL + jlc) * PW;
                    const float *ph = col + (V::XH + jhc - 8) * PW;
                    int up = 0, dn = 0;
                    static_range<0, 8>([&](auto I) NL_INL {
                        up += (pl[decltype(I)::value * PW] < wi.Lp) ? 1 : 0;
                        dn += (ph[decltype(I)::value * PW] > wi.Hm) ? 1 : 0;
                    });
                    if (more) {
                        jl = jlc + up;
                        jh = jhc - dn;
                        // a pointer at the inner end of its column: the shape test below hands the pixel over
                        more = (up == 8 && jl <= KL - 8) || (dn == 8 && jh >= 8);
                    }
                }
                // the clamped copy's moments: clamped ranks [a, jl) and local [jh, bl), unclamped the rest
                const int jlc = min(jl, KL - 4), jhc = max(jh, 4);
                const int gj = jlc >> 2, gh = jhc >> 2;
                float dp = 0.0f, qp = 0.0f;
                {
                    const float *pl = col + (V::XL + jlc) * PW;
                    const float *ph = col + (V::XH + jhc - 4) * PW;
                    const int nlp = 4 - (jlc & 3), nhp = jhc & 3;
                    static_range<0, 4>([&](auto I) NL_INL {
                        constexpr int i = decltype(I)::value;
                        const float e = (i < nlp) ? pl[i * PW] - c : 0.0f;
                        dp += e; qp = __builtin_fmaf(e, e, qp);
                        const float f = (i >= 4 - nhp) ? ph[i * PW] - c : 0.0f;
                        dp += f; qp = __builtin_fmaf(f, f, qp);
                    });
                }
                const float tl1 = tab[(V::SL1 + gj + 1) * PW], tl2 = tab[(V::SL2 + gj + 1) * PW];
                const float th1 = tab[(V::SH1 + gh) * PW], th2 = tab[(V::SH2 + gh) * PW];
                const float n_lo = (float)(jl - a), n_hi = (float)(bl - jh);
                const float eL = wi.Lp - c, eH = wi.Hm - c;               // max(x, Lp) - c of a clamped sample
                const float dcl = n_lo * eL + n_hi * eH;
                const float qcl = n_lo * (eL * eL) + n_hi * (eH * eH);
                const float wd = (((d_fix + dp) + (tl1 + th1)) + dcl) * inv_cnt;
                const float wa = (((q_fix + qp) + (tl2 + th2)) + qcl) * inv_cnt;
                const float wb = wd * wd;
                const float var_t = fmaxf(wa - wb, 0.0f);
                const float err_t = (kErrF + 8.0f) * kU * (wa + wb);
                // loosest clamp (Lm, Hp): first-order bound, see stack_fast.hip -- the counts are exact here
                float var_l;
                {
                    const float dL = (wi.Lp - wi.Lm) * (1.0f + 2.0f * kU), dH = (wi.Hp - wi.Hm) * (1.0f + 2.0f * kU);
                    const float ybar = c + wd;
                    // ybar is off by <= gamma_R mean|y-c| <= gamma_R sqrt(E[(y-c)^2]) plus its own rounding
                    const float slop = (float)(L::ROUNDINGS + 8) * kU * 1.01f * __builtin_amdgcn_sqrtf(wa) + 4.0f * kU * fabsf(ybar) + 1.0e-30f;
                    const float gL = fmaxf(ybar - wi.Lp, 0.0f) + slop, gH = fmaxf(wi.Hm - ybar, 0.0f) + slop;
                    const float corr = (n_lo * (dL * (2.0f * gL + dL)) + n_hi * (dH * (2.0f * gH + dH))) * inv_cnt;
                    var_l = var_t + ((corr == corr) ? corr * 1.001f : 0.0f);
                }
                // the clamps must stay inside the columns (and the ranks between the columns inside the clamps)
                const bool shape_ok = jl <= KL - 8 && jh >= 8 && x_in_lo >= wi.Lp && x_in_hi <= wi.Hm;
                if (inner && !shape_ok && role == 0) NL_STAT(5, 1);
                wi.finish_round(var_t, err_t, var_l, err_t, eps_r, e_m, shape_ok, inner, bail);
            }
            if (active && bail && role == 0) NL_STAT(wi.guard > 100 ? 7 : 6, 1);
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

        // ---- certain (c1, d1) and possible (c2, d2) clips among the CR outermost survivors per side ----
        int c1 = 0, c2 = 0, d1 = 0, d2 = 0;
        static_range<0, CR>([&](auto I) NL_INL {
            constexpr int i = decltype(I)::value;
            c1 += (xl[i] < lo_min) ? 1 : 0;
            c2 += (xl[i] < lo_max) ? 1 : 0;
            d1 += (xh[i] > hi_max) ? 1 : 0;
            d2 += (xh[i] > hi_min) ? 1 : 0;
        });
        if (active && role == 0) {
            if (c2 >= CR) NL_STAT(1, 1);
            else if (d2 >= CR) NL_STAT(2, 1);
            else if (a + c2 >= L::ZLC) NL_STAT(3, 1);
            else if (b - d2 <= NTOP - L::ZHC) NL_STAT(4, 1);
        }
        if (active && (c2 >= CR || d2 >= CR || a + c2 >= L::ZLC || b - d2 <= NTOP - L::ZHC)) {
            to_generic = true;                 // more clips than the columns hold: generic pass, from scratch
            active = false;
        }
        if (active) {
            bail |= (c1 != c2) || (d1 != d2) || (lo_max > hi_min && (c1 + d1) > 0);
            if (bail) {
                to_exact = true;
                active = false;
                if (p.nrounds && role == 0) p.nrounds[pix] = (unsigned char)min(rnd, kBoundRounds);
            } else {
                // the thresholds of a decided round go on record (winsorized passes: should the pixel turn undecidable
                // later, its replay skips the winsorization loop of this round, stack_fast_sigma_impl.hpp; decision pass
                // of a weighted stack, FastArgs::record_only: the replay only permutes)
                if (p.bounds) {
                    if (rnd < kBoundRounds && role == 0)
                        p.bounds[(size_t)rnd * (size_t)p.npix + (size_t)pix] = make_float2(lo_max, hi_min);
                    rnd++;
                }
                c_lo += c1;
                c_hi += d1;
                a += c1;
                b -= d1;
                if ((c1 + d1) == 0 || (b - a) <= 1) {      // stack.go:427-430: the mean BEFORE this pass
                    res = m;
                    active = false;
                }
            }
        }
    }

    // one lane per pixel reports
    const bool rec = q.record_only != 0;
    if (rec && on && role == 0) p.nrounds[pix] = (unsigned char)(to_generic ? 0 : min(rnd, kBoundRounds));
    const bool rep = on && role == 0 && !rec;
    if (rep && !to_generic && !to_exact) {
        NL_STORE_RESULT(&p.out[pix], res);
        c_lo_total += c_lo;
        c_hi_total += c_hi;
    }
    {
        const unsigned long long gm = __ballot(rep && to_generic);
        if (gm) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(q.gen_count, (unsigned)__popcll(gm));
            base = __shfl(base, 0, 64);
            const unsigned slot = base + (unsigned)__popcll(gm & ((1ull << lane) - 1ull));
            if (rep && to_generic && slot < q.gen_capacity) q.gen_list[slot] = (unsigned)pix;
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

#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c_lo_total += __shfl_xor(c_lo_total, o, 64);
        c_hi_total += __shfl_xor(c_hi_total, o, 64);
    }
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
    if (lane == 0) { NL_STAT(6, __builtin_readcyclecounter() - exp_t0); NL_STAT(7, 1); }
#endif
    if constexpr (L::PACK) {
        if (lane == 0) {                                   // (the one wave of the rounds phase)
            unsigned long long *slot = p.partial + 2 * (size_t)(blk % kClipSlots);
            if (c_lo_total) atomicAdd(slot + 0, (unsigned long long)c_lo_total);
            if (c_hi_total) atomicAdd(slot + 1, (unsigned long long)c_hi_total);
        }
    } else {
        if (lane == 0) { s_lo[threadIdx.x >> 6] = c_lo_total; s_hi[threadIdx.x >> 6] = c_hi_total; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t_lo = 0, t_hi = 0;
            for (int w = 0; w < L::BLOCK / 64; w++) { t_lo += s_lo[w]; t_hi += s_hi[w]; }
            unsigned long long *slot = p.partial + 2 * (size_t)(blk % kClipSlots);
            if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
            if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
        }
    }
    };   // ---- end of the rounds phase ----

    if constexpr (PHASE == 3) {
        // ---- persistent workgroups: block k of this workgroup is blockIdx.x + k * gridDim.x.  No barrier: wave (k mod 4)
        // runs the rounds of block k AFTER it has sorted its share of block k + 1 -- by then the other waves' rows of
        // block k have long been there, and its own lag (one rounds phase in four blocks, every wave in turn) never makes
        // another wave wait: nobody idles, nobody retires.  Two column buffers (a block's rows are overwritten two blocks
        // later, when its rounds are long over); the flags below only make that certain.
        // Blocks are handed out in order through a device counter (FastArgs::ticket): with a fixed stride per workgroup the
        // workgroups drift apart over hundreds of trips and the 512 frames are read at ever more scattered places.
        if (threadIdx.x < 2) { s_done[threadIdx.x] = 0u; s_freed[threadIdx.x] = 0u; }
        if (threadIdx.x == 0) { s_blk[0] = atomicAdd(q.ticket, 1u); s_seq = 1u; }
        __syncthreads();
        // All workgroups start together and every block takes the same time: without a stagger the three waves of a SIMD
        // would gather at the same time and sort at the same time for the whole launch -- nothing to hide the loads behind.
        // The k-th third of the grid (the dispatcher fills the CUs once per third) starts a third of a block's time later.
        {
            // (HW_ID[3:0]: the wave's slot on its SIMD)
            const int third = (int)((__builtin_amdgcn_s_getreg(63492) & 15u) % 3u);
            for (int i = 0; i < NL_MLZ_STAGGER * third; i++) __builtin_amdgcn_s_sleep(127);
        }
        // (a real call: inlined, the rounds phase's loop invariants -- lane masks, addresses, constants -- are kept in registers
        // across the sorting phase of every trip, which has none to spare: 55 spilled registers instead of 4, their reloads
        // on the waves' critical path)
        auto rounds_call = [&](const int64_t rblk, float *const rcol, float *const rtab) NL_INL {
            rounds_phase(rblk, rcol, rtab);
        };
        const int wave = (int)(threadIdx.x >> 6);
        const int64_t nblk = (p.npix + PW - 1) / PW;
        float *const tabbase = lds + 2 * V::ROWS * PW;
        int64_t prev_blk = 0;
        for (int k = 0;; k++) {
            mlz_spin_until(&s_seq, (unsigned)k + 1u);      // (wave 0 published this trip's block while sorting the last one)
            const int64_t blk = (int64_t)__builtin_amdgcn_readfirstlane((int)s_blk[k & 3]);
            const bool has = blk < nblk;                   // (the same for every wave of the workgroup)
            unsigned next_ticket = 0u;
            if (has && threadIdx.x == 0) next_ticket = atomicAdd(q.ticket, 1u);     // (returns behind the gather's loads)
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
            const unsigned long long tt0 = __builtin_readcyclecounter();
#endif
            if (has) {
                float *const buf = lds + (k & 1) * (V::ROWS * PW);
                sorting_phase(blk, buf, [&]() NL_INL {
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
                    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
                    if (k >= 2) mlz_spin_until(&s_freed[k & 1], (unsigned)(k >> 1));      // rounds of block k - 2 are over
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
                    if (threadIdx.x == 0) NL_STAT(3, __builtin_readcyclecounter() - ts0);
#endif
                });
                lds_settle();
                if (lane == 0) __hip_atomic_fetch_add(&s_done[k & 1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (threadIdx.x == 0) {
                    s_blk[(k + 1) & 3] = next_ticket;
                    __hip_atomic_store(&s_seq, (unsigned)k + 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
                if (threadIdx.x == 0) { NL_STAT(5, __builtin_readcyclecounter() - tt0); NL_STAT(4, 1); }
#endif
            }
            const int j = k - 1;
            if (j >= 0 && (j & 3) == wave) {
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
                const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
                mlz_spin_until(&s_done[j & 1], 4u * (unsigned)((j >> 1) + 1));             // every wave's rows of block j
                if (j >= 1) mlz_spin_until(&s_freed[(j - 1) & 1], (unsigned)(((j - 1) >> 1) + 1));      // the table area is free
#if defined(NL_ROUND_STATS) && defined(NL_MLZ_EXP_TIMING)
                if (lane == 0) NL_STAT(2, __builtin_readcyclecounter() - ts1);
#endif
                rounds_call(prev_blk, lds + (j & 1) * (V::ROWS * PW), tabbase);
                lds_settle();
                if (lane == 0) __hip_atomic_fetch_add(&s_freed[j & 1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (!has) break;
            prev_blk = blk;
        }
        return;
    } else {
    if constexpr (PHASE == 2) {
        // ---- the columns the sorting kernel left: rows of 64 pixels, back to their places in LDS ----
        const float *src = q.cols + (size_t)blockIdx.x * (size_t)(SP::N * PW) + lane;
        static_range<0, SP::N>([&](auto G) NL_INL {
            constexpr int g = decltype(G)::value;
            lds[SP::row(g) * PW + lane] = src[g * PW];
        });
        lds_settle();
    } else {
        sorting_phase((int64_t)blockIdx.x, lds, []() NL_INL {});
    }
    if constexpr (PHASE == 1) {
        // the columns, the window and the scalars of the workgroup's 64 pixels: 256-byte rows, every wave its share
        __syncthreads();
        float *dst = q.cols + (size_t)blockIdx.x * (size_t)(SP::N * PW) + lane;
        static_range<0, (SP::N + 3) / 4>([&](auto G) NL_INL {
            const int g = 4 * decltype(G)::value + (int)(threadIdx.x >> 6);
            if (g < SP::N) dst[g * PW] = lds[SP::row_rt(g) * PW + lane];
        });
        return;
    } else if constexpr (PHASE == 2) {
        // (one wave: nothing to meet)
    } else if constexpr (L::PACK) {
        __syncthreads();
#ifdef NL_MLZ_EXP_SORTONLY
        if (p.npix > 0) return;                            // (timing experiment: the sorting phase alone)
#endif
        if ((int)(threadIdx.x >> 6) != (int)(blockIdx.x % (L::BLOCK / 64))) return;
#ifdef NL_MLZ_EXP_PRIO
        __builtin_amdgcn_s_setprio(NL_MLZ_EXP_PRIO);
#endif
#ifdef NL_MLZ_EXP_SLEEP
        if (p.npix > 0) {                                  // (timing experiment: the rounds wave only holds its slot)
            for (int i = 0; i < NL_MLZ_EXP_SLEEP; i++) __builtin_amdgcn_s_sleep(127);
            return;
        }
#endif
        // (s_setprio 3 for this wave -- it holds the workgroup's LDS -- measured 0.5 % slower, two interleaved runs)
    } else {
        lds_settle();
    }
    rounds_phase((int64_t)blockIdx.x, lds, lds);
    }
}


// launches the kernel of frame-count class `ntop` if this translation unit instantiates it
template <int LPP, int... NTOPS>
static bool launch_mlz_classes(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream,
                               std::integer_sequence<int, NTOPS...>)
{
    bool done = false;
    auto one = [&](auto C) {
        constexpr int NTOP = decltype(C)::value;
        if (done || ntop != NTOP) return;
        using L = MlzLayout<LPP, false, NTOP>;
        using LW = MlzLayout<LPP, true, NTOP>;
        if (winsor) hipLaunchKernelGGL((stack_sigma_mlz_kernel<LPP, true, NTOP>), dim3((unsigned)((args.npix + LW::PW - 1) / LW::PW)), dim3(LW::BLOCK), 0, stream, args, f);
        else if constexpr (L::SELECT) {
            const dim3 grid((unsigned)((args.npix + L::PW - 1) / L::PW));
#ifdef NL_EXPERIMENTS
            // (both measured slower than the one-kernel pass, DESIGN.md section 5n: instantiated in the experiments build only)
            if (f.cols) {
                // split pass: the sorting kernel's workgroups retire as a whole (in the one-kernel pass three of a
                // workgroup's four wave slots idle while its fourth wave runs the rounds), then one wave per 64 pixels
                hipLaunchKernelGGL((stack_sigma_mlz_kernel<LPP, false, NTOP, 1>), grid, dim3(L::BLOCK), 0, stream, args, f);
                hipLaunchKernelGGL((stack_sigma_mlz_kernel<LPP, false, NTOP, 2>), grid, dim3(64), 0, stream, args, f);
            } else if (f.persistent) {
                // persistent workgroups, three per CU (168 registers: three waves per SIMD), each looping over blocks of 64 pixels
                static const int cus = [] {
                    int dev = 0, n = 0;
                    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
                    return n;
                }();
                static const int per_cu = [] { const char *e = getenv("NL_MLZ_WGS_PER_CU"); const int v = e ? atoi(e) : 3; return v >= 1 && v <= 8 ? v : 3; }();      // (experiments)
                const unsigned wgs = (unsigned)(per_cu * cus);
                hipLaunchKernelGGL((stack_sigma_mlz_kernel<LPP, false, NTOP, 3>), dim3(grid.x < wgs ? grid.x : wgs), dim3(L::BLOCK), 0, stream, args, f);
            } else
#endif
            {
                hipLaunchKernelGGL((stack_sigma_mlz_kernel<LPP, false, NTOP>), grid, dim3(L::BLOCK), 0, stream, args, f);
            }
        }
        else        hipLaunchKernelGGL((stack_sigma_mlz_kernel<LPP, false, NTOP>), dim3((unsigned)((args.npix + L::PW - 1) / L::PW)), dim3(L::BLOCK), 0, stream, args, f);
        done = true;
    };
    (one(std::integral_constant<int, NTOPS>{}), ...);
    return done;
}

}  // namespace nl
