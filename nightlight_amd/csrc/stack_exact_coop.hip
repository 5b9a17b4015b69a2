// stack_exact_coop.hip -- bit-exact StackSigma / StackWinsorSigma replay, one WAVEFRONT per pixel.
//
// The register-resident kernels hand ~1e-4 of the pixels (a sample inside the
// few-ulp clip window) to an exact replay.  With so few pixels the
// one-pixel-per-lane kernel of stack_exact.hip is pure latency (a handful of
// active lanes crawling through divergent loops); here the 64 lanes of a wave
// cooperate on ONE pixel and still execute the reference's algorithm in the
// reference's order, so every bit and both counters are unchanged:
//   gather      stack.go:380-387   64 frames per step, order-preserving compaction (ballot + popcount)
//   quickselect qsort.go:94-126    a whole Hoare partition pass at once: the misplaced elements of
//                                  both sides are listed with ballot + popcount and the pass's
//                                  swaps are done in parallel (see coop_select)
//   mean/stddev stats.go:246-261   the fp32 sums stay sequential: a DPP wave-shift add chain, 64
//                                  elements per step; differences and squares are computed 64 at a time
//   winsorize   stack.go:646-672   the copy is clamped 64 samples at a time (ballot counts `changed`),
//                                  its mean / stddev are the same sequential sums
//   clip        stack.go:411-424   swap-with-last, same visiting order, clean stretches skipped 64 at a time
// The pixel's column lives in LDS as a plain array (n_frames floats per wave;
// the winsorized variant keeps its clamped copy in a second one).
#include "stack_kernels.h"
#include <stdint.h>
#include <stdlib.h>

namespace nl {

#ifdef NL_PROBE
__device__ unsigned long long nl_probe_cycles[8];
#define NL_T0() nl_t = (long long)__builtin_readcyclecounter()
#define NL_T(slot) do { const long long nl_n = (long long)__builtin_readcyclecounter(); nl_acc[slot] += nl_n - nl_t; nl_t = nl_n; } while (0)
#define NL_TDECL() long long nl_t = 0, nl_acc[5] = {0, 0, 0, 0, 0}; const long long nl_c0 = (long long)__builtin_readcyclecounter(), nl_r0 = (long long)__builtin_amdgcn_s_memrealtime()
#define NL_TFLUSH() do { if (threadIdx.x == 0) { for (int q = 0; q < 5; q++) atomicAdd(&nl_probe_cycles[q], (unsigned long long)nl_acc[q]); \
        atomicAdd(&nl_probe_cycles[5], (unsigned long long)((long long)__builtin_readcyclecounter() - nl_c0)); \
        const unsigned long long nl_w = (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - nl_r0); \
        atomicAdd(&nl_probe_cycles[6], nl_w); atomicMax(&nl_probe_cycles[7], nl_w); } } while (0)
#else
#define NL_T0() do {} while (0)
#define NL_T(slot) do {} while (0)
#define NL_TDECL() do {} while (0)
#define NL_TFLUSH() do {} while (0)
#endif

namespace {

__device__ __forceinline__ float sqrt_like_go(float x)      // stats.go:259
{
    return (float)__builtin_sqrt((double)x);
}

__device__ __forceinline__ unsigned long long ballot64(bool p)      // the compare's own wave mask (HIP's __ballot goes
{                                                                    // through an integer: v_cndmask + v_cmp_ne on top)
    return __builtin_amdgcn_ballot_w64(p);
}

__device__ __forceinline__ int below64(unsigned long long m)         // set bits of m below this lane: two v_mbcnt
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ int first_lane(unsigned long long m)      // lowest set bit, -1 for an empty mask (s_ff1's own answer;
{                                                                    // __builtin_ffsll wraps it in a compare and a select)
    int r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r;
}

__device__ __forceinline__ void lds_fence()
{
    __syncthreads();      // single-wave workgroup: orders LDS writes before later reads
}

// sequential fp32 sum of t[0..n) in index order; elem = per-lane slice loader, which must
// deliver +0.0f past n (adding +0.0f leaves a running sum unchanged bit for bit: the sum
// starts at +0.0f and can therefore never be -0.0f).
//
// 64 elements per step: lane l holds x[l]; "s[l] = s[l-1] + x[l]" is issued 63 times on
// all lanes with a DPP wave shift (lane 0, whose source is out of range, is left alone).
// After step t lanes 0..t hold their final prefix sums -- re-computing a final value from a
// final neighbour gives the same bits -- so lane 63 ends with the chunk's sequential sum,
// one VALU instruction per element.
__device__ __forceinline__ float chain64(float carry, float x)
{
    float s = (threadIdx.x == 0) ? carry + x : x;
#define NL_STEP "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
#define NL_STEP8 NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP
    asm volatile("s_nop 1\n\t" NL_STEP8 NL_STEP8 NL_STEP8 NL_STEP8 NL_STEP8 NL_STEP8 NL_STEP8
                 NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP
                 : "+v"(s) : "v"(x));
#undef NL_STEP8
#undef NL_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
}

template <class F>
__device__ __forceinline__ float seq_sum(int n, F &&elem)
{
    float s = 0.0f;
    const int lane = threadIdx.x;
    for (int base = 0; base < n; base += 64) s = chain64(s, elem(base + lane));
    return s;
}

// Two independent sequential sums at once (the weighted mean's numerator and denominator, stack.go:514-522):
// one occupies lanes 0..31, the other lanes 32..63, 32 elements of each per step, and they swap halves from step
// to step so that the carry is a single wave rotation (lane 31 -> 32, lane 63 -> 0).  Inside a half the chain is
// two 16-lane rows: 15 row shifts, lane 15 broadcast into the next row, 15 row shifts -- 32 VALU instructions per
// 32 + 32 elements instead of 2 x 63 per 64 + 64.  Lanes that are not yet final hold anything: a final value only
// ever comes from a final neighbour (see chain64).  `s`: the previous step's register (0.0f before the first).
__device__ __forceinline__ float chain32x2(float s, float x)
{
#define NL_ROW(mask) "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:" mask " bank_mask:0xf\n\ts_nop 1\n\t"
#define NL_ROW15(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) \
                       NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask) NL_ROW(mask)
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0x5 bank_mask:0x1\n\ts_nop 1\n\t"
                 NL_ROW15("0x5")
                 "v_add_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0x1\n\ts_nop 1\n\t"
                 NL_ROW15("0xa")
                 : "+v"(s) : "v"(x));
#undef NL_ROW15
#undef NL_ROW
    return s;
}

// sums of ea(i) and eb(i), i = 0 .. n-1, each in index order; the loaders deliver +0.0f past n
template <class FA, class FB>
__device__ __forceinline__ void seq_sum2(int n, FA &&ea, FB &&eb, float &sum_a, float &sum_b)
{
    const int lane = threadIdx.x;
    float s = 0.0f;
    int c = 0;
    for (int base = 0; base < n; base += 32, c++) {
        const int i = base + (lane & 31);
        const bool first = (((lane >> 5) ^ c) & 1) == 0;     // the half that holds the first sum in this step
        s = chain32x2(s, first ? ea(i) : eb(i));
    }
    const float e31 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 31));
    const float e63 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
    sum_a = (c & 1) ? e31 : e63;                             // the last step was c - 1
    sum_b = (c & 1) ? e63 : e31;
}

// The partition passes of a select whose range [left, right] has shrunk to at most 63 elements, in REGISTERS:
// lane i holds a[left + i]; a pass is two ballots, the candidate tables through ds_permute (lane t receives the
// position of the t-th candidate from the left / from the right; lanes that have nothing to send aim at lane 63,
// which holds no element) and the swaps through ds_bpermute / ds_permute -- no LDS traffic, no barrier.  Same
// swaps, same pointers as coop_select below (k is 1-based inside [left, right]); the range is written back at
// the end.  About a third of the latency of the LDS form per pass, and most passes of a select are this small.
__device__ float coop_select_small(float *a, int left, int right, int k)
{
    const int lane = threadIdx.x;
    int lo = 0, hi = __builtin_amdgcn_readfirstlane(right - left);
    const int target = __builtin_amdgcn_readfirstlane(k) - 1;         // position of the wanted element: lo + k - 1 never changes
    const bool mine = lane <= hi;
    float x = mine ? a[left + lane] : 0.0f;
    constexpr int kTrash = 63 * 4;
    // With eight waves per SIMD the replay is bound by the ISSUE of scalar and vector instructions alike (about 40 of
    // each per pass at first; cycle probes, tools/coop_probe.py), so a pass is written for few of both: the
    // classification lives in wave masks (a v_cmp IS the ballot; HIP's __ballot goes through an integer), ranks come
    // from v_mbcnt, the range mask from one s_bfm, nothing in the loop body branches, and who swaps follows from
    // the ranks alone, without tables of candidate positions:
    //   the L-candidate at position p with rank t (t candidates below it) swaps  <=>  L_t < R_t
    //        <=>  at least t + 1 R-candidates lie above p;
    //   the R-candidate at position q with rank u (u candidates above it) swaps  <=>  at least u + 1 L-candidates lie below q.
    // Each side then scatters its values to the lane of their rank (ds_permute) and the swapping candidates of the
    // other side fetch the value of their own rank (ds_bpermute): two dependent LDS-crossbar trips instead of three.
    // The pass ends at r = max(R_s, L_{s-1}): the position of the candidate whose rank is s (s - 1), by s_ff1 over
    // the classification mask (-1 when there is none, as the formula wants).
    while (lo < hi) {
        const int pm = (lo + hi) >> 1;                       // (left + right) >> 1, relative to left
        const float pivot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), pm));
        unsigned long long in;                               // lanes lo .. hi (hi <= 62)
        asm("s_bfm_b64 %0, %1, %2" : "=s"(in) : "s"(hi - lo + 1), "s"(lo));
        const unsigned long long ml = ballot64(x >= pivot) & in;
        const unsigned long long mr = ballot64(x <= pivot) & in;
        const bool isl = __builtin_amdgcn_inverse_ballot_w64(ml), isr = __builtin_amdgcn_inverse_ballot_w64(mr);
        const int nr = __popcll(mr);
        const int rl = below64(ml);                          // L-candidates below this position
        const int ra = nr - below64(mr) - (isr ? 1 : 0);     // R-candidates above this position
        const unsigned long long sl = ml & ballot64(ra > rl);                         // the swapping L-candidates (rank rl) ...
        const unsigned long long sr = mr & ballot64(rl > ra);                         // ... and R-candidates (rank ra); never both
        const int s_cnt = __popcll(sl);
        const int val_r = __builtin_amdgcn_ds_permute(isr ? ra * 4 : kTrash, __float_as_int(x));      // lane t: a[R_t]
        const int val_l = __builtin_amdgcn_ds_permute(isl ? rl * 4 : kTrash, __float_as_int(x));      // lane t: a[L_t]
        const int from_r = __builtin_amdgcn_ds_bpermute(rl * 4, val_r);
        const int from_l = __builtin_amdgcn_ds_bpermute(ra * 4, val_l);
        const int r_s = first_lane(mr & ballot64(ra == s_cnt));                       // R_s, -1 if there is none
        const int l_p = first_lane(ml & ballot64(rl == s_cnt - 1));                   // L_{s-1}
        int xi = __float_as_int(x);
        xi = __builtin_amdgcn_inverse_ballot_w64(sl) ? from_r : xi;
        xi = __builtin_amdgcn_inverse_ballot_w64(sr) ? from_l : xi;
        x = __int_as_float(xi);
        const int r = max(r_s, l_p);
        if (target <= r) hi = r; else lo = r + 1;            // k <= r - lo + 1, with k = target - lo + 1
    }
    if (mine) a[left + lane] = x;
    const float res = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lo));
    lds_fence();
    return res;
}

// The same for a range of 64 ... 128 elements, two per lane: x0 = positions 0 .. 63 of the range, x1 = 64 .. 127.
// Ranks run through both registers (v_mbcnt takes the first register's count as its base); a pass swaps at most
// 64 pairs (2 s <= 128) and only ranks below s are ever fetched, so one 64-lane rank table per side is enough:
// candidates of rank >= 63 aim at the trash lane with everything that is no candidate.  (s = 64 needs the rank-63
// entry: possible only with all 128 positions in range -- the function then returns false before it has changed
// anything and the caller runs that one pass through LDS.)  Leaves through coop_select_small once fewer than 64
// elements are in range.  *res = the selected element; [left, right] is updated by nothing: the result is final.
__device__ bool coop_select_mid(float *a, int left, int right, int k, float *res)
{
    const int lane = threadIdx.x;
    int lo = 0, hi = __builtin_amdgcn_readfirstlane(right - left);                    // 63 ... 127
    const int top = hi;
    const int target = __builtin_amdgcn_readfirstlane(k) - 1;
    float x0 = a[left + lane];
    float x1 = lane + 64 <= top ? a[left + 64 + lane] : 0.0f;
    constexpr int kTrashLane = 63;
    bool complete = true;
    while (hi - lo >= 63) {                                  // (then lo <= 64, hi >= 63)
        const int pm = (lo + hi) >> 1;
        const float xs = (pm & 64) ? x1 : x0;
        const float pivot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), pm & 63));
        const unsigned long long in0 = lo < 64 ? ~0ull << lo : 0ull;               // (lo = 64: the 64 elements of x1)
        const unsigned long long in1 = hi >= 64 ? ~0ull >> (127 - hi) : 0ull;
        const unsigned long long ml0 = ballot64(x0 >= pivot) & in0, ml1 = ballot64(x1 >= pivot) & in1;
        const unsigned long long mr0 = ballot64(x0 <= pivot) & in0, mr1 = ballot64(x1 <= pivot) & in1;
        const bool isl0 = __builtin_amdgcn_inverse_ballot_w64(ml0), isl1 = __builtin_amdgcn_inverse_ballot_w64(ml1);
        const bool isr0 = __builtin_amdgcn_inverse_ballot_w64(mr0), isr1 = __builtin_amdgcn_inverse_ballot_w64(mr1);
        const int nl0 = __popcll(ml0), nr1 = __popcll(mr1), nr = nr1 + (int)__popcll(mr0);
        // L-candidates below / R-candidates above each position
        const int rl0 = below64(ml0);
        const int rl1 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ml1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml1, (unsigned)nl0));
        const int ra1 = nr1 - below64(mr1) - (isr1 ? 1 : 0);
        const int ra0 = nr - below64(mr0) - (isr0 ? 1 : 0);
        const unsigned long long sl0 = ml0 & ballot64(ra0 > rl0), sl1 = ml1 & ballot64(ra1 > rl1);
        const unsigned long long sr0 = mr0 & ballot64(rl0 > ra0), sr1 = mr1 & ballot64(rl1 > ra1);
        const int s_cnt = (int)__popcll(sl0) + (int)__popcll(sl1);
        if (s_cnt > 63) { complete = false; break; }
        const int al0 = min(rl0, kTrashLane) * 4, al1 = min(rl1, kTrashLane) * 4;    // rank as a crossbar address
        const int ar0 = min(ra0, kTrashLane) * 4, ar1 = min(ra1, kTrashLane) * 4;
        const int vr_a = __builtin_amdgcn_ds_permute(isr0 ? ar0 : kTrashLane * 4, __float_as_int(x0));
        const int vr_b = __builtin_amdgcn_ds_permute(isr1 ? ar1 : kTrashLane * 4, __float_as_int(x1));
        const int vl_a = __builtin_amdgcn_ds_permute(isl0 ? al0 : kTrashLane * 4, __float_as_int(x0));
        const int vl_b = __builtin_amdgcn_ds_permute(isl1 ? al1 : kTrashLane * 4, __float_as_int(x1));
        const int val_r = lane < nr1 ? vr_b : vr_a;          // lane t: a[R_t] (the candidates of x1 come first from the right)
        const int val_l = lane < nl0 ? vl_a : vl_b;          // lane t: a[L_t]
        const int fr0 = __builtin_amdgcn_ds_bpermute(al0, val_r), fr1 = __builtin_amdgcn_ds_bpermute(al1, val_r);
        const int fl0 = __builtin_amdgcn_ds_bpermute(ar0, val_l), fl1 = __builtin_amdgcn_ds_bpermute(ar1, val_l);
        // r = max(R_s, L_{s-1})
        const int rs0 = first_lane(mr0 & ballot64(ra0 == s_cnt));
        const int rs1 = first_lane(mr1 & ballot64(ra1 == s_cnt));
        const int lp0 = first_lane(ml0 & ballot64(rl0 == s_cnt - 1));
        const int lp1 = first_lane(ml1 & ballot64(rl1 == s_cnt - 1));
        const int r = max(max(rs0, rs1 < 0 ? -1 : rs1 + 64), max(lp0, lp1 < 0 ? -1 : lp1 + 64));
        int xi0 = __float_as_int(x0), xi1 = __float_as_int(x1);
        xi0 = __builtin_amdgcn_inverse_ballot_w64(sl0) ? fr0 : xi0;
        xi0 = __builtin_amdgcn_inverse_ballot_w64(sr0) ? fl0 : xi0;
        xi1 = __builtin_amdgcn_inverse_ballot_w64(sl1) ? fr1 : xi1;
        xi1 = __builtin_amdgcn_inverse_ballot_w64(sr1) ? fl1 : xi1;
        x0 = __int_as_float(xi0);
        x1 = __int_as_float(xi1);
        if (target <= r) hi = r; else lo = r + 1;
    }
    if (!complete) return false;                             // (first pass of a 128-element range: nothing has moved)
    a[left + lane] = x0;
    if (lane + 64 <= top) a[left + 64 + lane] = x1;
    lds_fence();
    if (lo < hi) *res = coop_select_small(a, left + lo, left + hi, target - lo + 1);
    else         *res = a[left + lo];
    return true;
}

// qsort.go:94-126 on a[0..n), k 1-based; all control values are wave-uniform.
//
// One Hoare partition pass (qsort.go:100-114) done by the whole wave at once.
// Sequentially, l stops at the misplaced elements from the left (a >= pivot), r
// at those from the right (a <= pivot), they are swapped pairwise and the pass
// ends when the pointers meet.  Let L_0 < L_1 < ... be the positions with
// a >= pivot and R_0 > R_1 > ... those with a <= pivot, both in the ORIGINAL
// array.  By induction the i-th swap is exactly (L_i, R_i) as long as
// L_i < R_i: the stretch between the pointers is still unmodified, and the
// swapped-in values stop the opposite pointer no earlier than its own next
// candidate (l_{i+1} = min(L_{i+1}, R_i), r_{i+1} = max(R_{i+1}, L_i)).  With
// s = #{i : L_i < R_i} the pass performs the swaps i < s -- disjoint positions,
// so they can be done in parallel -- and ends with r = max(R_s, L_{s-1}).
// The resulting array is identical to the sequential one, element for element.
__device__ float coop_select(float *a, unsigned short *lpos, unsigned short *rfwd, int n, int k)
{
    const int lane = threadIdx.x;
    int left = 0, right = n - 1;
    while (left < right) {
        if (right - left < 63) {
            return coop_select_small(a, left, right, k);
        }
        if (right - left <= 127) {
            float res;
            if (coop_select_mid(a, left, right, k, &res)) return res;
        }
        const float pivot = a[(left + right) >> 1];
        // classify, and list the misplaced positions in scan order
        int nl = 0, nr = 0;
        for (int base = left; base <= right; base += 64) {
            const int idx = base + lane;
            const bool in = idx <= right;
            const float x = in ? a[idx] : 0.0f;
            const bool isl = in && x >= pivot;
            const bool isr = in && x <= pivot;
            const unsigned long long ml = ballot64(isl), mr = ballot64(isr);
            if (isl) lpos[nl + below64(ml)] = (unsigned short)idx;         // (positions < 65536 by coop_supported: 16 bits, half the LDS)
            if (isr) rfwd[nr + below64(mr)] = (unsigned short)idx;         // ascending; R_i = rfwd[nr-1-i]
            nl += __popcll(ml);
            nr += __popcll(mr);
        }
        lds_fence();
        // s = number of leading pairs with L_i < R_i (the predicate is monotone in i)
        const int pairs = min(nl, nr);
        int s_cnt = 0;
        for (int base = 0; base < pairs; base += 64) {
            const int i = base + lane;
            const bool ok = i < pairs && (int)lpos[i] < (int)rfwd[nr - 1 - i];
            const unsigned long long m = ballot64(ok);
            s_cnt += __popcll(m);
            if (m != ~0ull) break;
        }
        // the pass's swaps, all at once
        for (int base = 0; base < s_cnt; base += 64) {
            const int i = base + lane;
            if (i < s_cnt) {
                const int pl = (int)lpos[i], pr = (int)rfwd[nr - 1 - i];
                const float xl = a[pl], xr = a[pr];
                a[pl] = xr;
                a[pr] = xl;
            }
        }
        const int r_next = s_cnt < nr ? (int)rfwd[nr - 1 - s_cnt] : -1;
        const int l_prev = s_cnt > 0 ? (int)lpos[s_cnt - 1] : -1;
        const int r = max(r_next, l_prev);
        lds_fence();
        const int offset = r - left + 1;
        if (k <= offset) {
            right = r;
        } else {
            left = r + 1;
            k -= offset;
        }
    }
    return a[left];
}

// qsort.go:68-82
__device__ float coop_select_median(float *a, unsigned short *lpos, unsigned short *rfwd, int n)
{
    const int k = (n >> 1) + 1;
    const float upper = coop_select(a, lpos, rfwd, n, k);
    if (n & 1) return upper;
    // max of a[0..k-2]
    const int lane = threadIdx.x;
    float lower = -__builtin_inff();
    for (int base = 0; base < k - 1; base += 64) {
        const int idx = base + lane;
        if (idx < k - 1) lower = fmaxf(lower, a[idx]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lower = fmaxf(lower, __shfl_xor(lower, o, 64));
    return 0.5f * (lower + upper);
}

}  // namespace

// W: weighted variants (stack.go:442-531, 710-829).  The weights live in their own column and
// follow only the clip swaps -- quickselect permutes the samples, NOT the weights
// (stack.go:487), and the weighted mean pairs them index by index all the same.
//
// GROUP = 4: a replay over the whole tile (no list) whose work items are four consecutive pixels.  A lane loads
// the samples of all four at once (16 bytes: the 128 frames x 4 pixels sit in eight registers per lane) and the
// pixels are replayed one after the other.  One pixel at a time fetched 13 x the algorithmic bytes (4 bytes out
// of every 64-byte sector, with 16 MB of lines in flight per XCD against 4 MB of L2) and the replay ran at the
// speed of those fetches.
// PF: chunks of 64 frames a work item holds in registers before it replays its pixels (GROUP = 4: as 16-byte loads of
// all four).  Round 5: deep weighted stacks took PF = 2 like everything else and fetched the frames beyond 128 four bytes
// at a time, once per pixel -- 13.7 x the algorithmic bytes at 512 frames, 3.4 TB/s of mostly unused sectors
// (profiles/r04_wsigma512_*).  With PF = 8 (257 ... 512 frames) every frame of the four pixels arrives in one 16-byte load
// per lane; the registers cost wave slots the LDS columns of those depths had taken already.  Dispatched for the
// winsorized replays only, see launch_coop.
// The kernel's body, workgroup `block` of `nblocks` with its LDS columns at `a`: stack_sigma_coop_kernel below is the whole grid;
// stack_tail_fused.hip runs it in the upper part of a grid whose lower workgroups are the generic pass.
template <bool WINSOR, bool W, int GROUP, int PF = 2>
__device__ __forceinline__ void coop_body(const StackArgs &p, float *a, const unsigned block, const unsigned nblocks)
{
    float *wz = a + p.n_frames;                   // winsorized copy (WINSOR only)
    float *wt = a + (WINSOR ? 2 : 1) * p.n_frames;          // weights (W only)
    unsigned short *lpos = reinterpret_cast<unsigned short *>(a + ((WINSOR ? 2 : 1) + (W ? 1 : 0)) * p.n_frames);   // partition scratch, 2 x n_frames x 16 bit
    unsigned short *rfwd = lpos + p.n_frames;
    const int lane = threadIdx.x;
    const int N = p.n_frames;
    int64_t limit = p.npix;
    if (p.list) {
        // atomic load: a generic pass on another stream may be appending (see snapshot_fb_list, fast_common.hpp)
        const unsigned cnt = __atomic_load_n(p.list_count, __ATOMIC_RELAXED);
        limit = cnt < p.list_capacity ? cnt : p.list_capacity;
    }
    long long c_lo = 0, c_hi = 0;
    NL_TDECL();

    int64_t first = 0;
    if (p.list && p.list_snap) {
        unsigned s = 0;
        if (lane == 0) {
            s = __atomic_load_n(p.list_snap, __ATOMIC_RELAXED);        // (a plain look first: one address, thousands of workgroups)
            if (s == 0u) {
                s = atomicCAS(p.list_snap, 0u, (unsigned)limit + 1u);
                if (s == 0u) s = (unsigned)limit + 1u;
            }
        }
        s = (unsigned)__shfl((int)s, 0, 64);
        const int64_t snap = min((int64_t)(s - 1u), limit);
        if (p.list_part == 0) limit = snap; else first = snap;
    }
    // Whole-tile replays (no list): a pixel's sample is 4 bytes of a 128-byte line that 31 neighbours share.
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, so with pixel = workgroup index the
    // neighbours sit on other XCDs and every line is fetched again and again (29.6 x the algorithmic bytes,
    // measured).  XCD x takes the x-th eighth of every sweep instead: neighbours run side by side under one L2.
    int64_t wg = block;
    if (!p.list && (nblocks & 7u) == 0u) wg = (int64_t)(block & 7u) * (nblocks >> 3) + (block >> 3);
    // The first 128 frames of a pixel are gathered up front (two loads in flight), its decided rounds and their
    // bounds come with them (lane r holds round r), the weights of those frames sit in registers.
    static_assert(PF == 2 || GROUP == 4, "deep register prefetch is for the whole-tile replays");
    const bool dense = p.list == nullptr;
    float wreg[PF] = {};
    if constexpr (W) {
#pragma unroll
        for (int c = 0; c < PF; c++) wreg[c] = c * 64 + lane < N ? p.weights[c * 64 + lane] : 0.0f;
    }
    if constexpr (GROUP > 1) limit = p.npix / GROUP;          // (never with a list; npix is a multiple of GROUP)
    for (int64_t item = first + wg; item < limit; item += nblocks) {
      float4 grp[PF];
      if constexpr (GROUP > 1) {
        static_assert(GROUP == 4, "one 16-byte load per lane and chunk");
#pragma unroll
        for (int c = 0; c < PF; c++) {
            const int k = min(c * 64 + lane, N - 1);          // (frames past the stack are masked below)
            grp[c] = *reinterpret_cast<const float4 *>(p.frames + item * GROUP + (int64_t)k * p.stride);
        }
      }
      for (int j = 0; j < GROUP; j++) {
        const int64_t pix = GROUP > 1 ? item * GROUP + j : (dense ? item : (int64_t)p.list[item]);
        const float *fr = p.frames + pix;
        const int64_t fstride = p.stride;
        float cur[PF];
#pragma unroll
        for (int c = 0; c < PF; c++) {
            const int k = c * 64 + lane;
            if constexpr (GROUP > 1) {
                const float x = j == 0 ? grp[c].x : j == 1 ? grp[c].y : j == 2 ? grp[c].z : grp[c].w;
                cur[c] = k < N ? x : __builtin_nanf("");
            } else {
                cur[c] = k < N ? fr[(int64_t)k * fstride] : __builtin_nanf("");
            }
        }
        const int decided = p.nrounds ? (int)p.nrounds[pix] : 0;
        float2 bd = make_float2(0.0f, 0.0f);                  // lane r: the bounds of round r
        if (decided > 0 && lane < kBoundRounds) bd = p.bounds[(size_t)lane * (size_t)p.npix + (size_t)pix];
        NL_T0();
        lds_fence();
        // ---- gather in frame order, NaN dropped (stack.go:380-387) ----
        int n = 0;
#pragma unroll
        for (int c = 0; c < PF; c++) {
            if (c * 64 < N) {
                const float x = cur[c];
                const bool valid = x == x;
                const unsigned long long m = ballot64(valid);
                const int pos = n + below64(m);
                if (valid) a[pos] = x;
                if (W && valid) wt[pos] = wreg[c];                    // stack.go:452-459
                n += __popcll(m);
            }
        }
        for (int base = PF * 64; base < N; base += 64) {
            const int k = base + lane;
            const float x = k < N ? fr[(int64_t)k * fstride] : __builtin_nanf("");
            const bool valid = x == x;
            const unsigned long long m = ballot64(valid);
            const int pos = n + below64(m);
            if (valid) a[pos] = x;
            if (W && valid) wt[pos] = p.weights[k];
            n += __popcll(m);
        }
        lds_fence();

        NL_T(0);
        float res = p.ref_loc;
        // StackArgs::bounds (weighted stacks with a decision pass; list replays of winsorized passes): the clip
        // bounds of this pixel's first `decided` rounds are on record -- those rounds only permute (the quickselect of
        // QSelectMedian) and clip.  An unweighted result is the mean of the LAST round, which is never on record.
        int rnd = 0;
        if (n > 0) {
            for (;;) {
                float lo, hi, mean = 0.0f;
                if (rnd < decided) {
                    (void)coop_select(a, lpos, rfwd, n, (n >> 1) + 1);      // qsort.go:70 (the even-n scan of :73-81 does not permute)
                    lds_fence();
                    NL_T(1);
                    lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd.x), rnd));
                    hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd.y), rnd));
                } else {
                const float median = coop_select_median(a, lpos, rfwd, n);
                lds_fence();
                // stats.go:246-261
                const float fn = (float)n;
                const float s = seq_sum(n, [&](int i) { return i < n ? a[i] : 0.0f; });
                mean = s / fn;
                const float vs = seq_sum(n, [&](int i) {
                    const float d = (i < n ? a[i] : mean) - mean;
                    return i < n ? d * d : 0.0f;
                });
                const float var = vs / fn;
                float sd = sqrt_like_go(var);
                if constexpr (WINSOR) {
                    // stack.go:646-672: clamp a copy to median -/+ 1.5 sd, sd = 1.134 * stddev(copy),
                    // until no sample moved or sd changed by <= 0.05 %
                    for (int base = 0; base < n; base += 64)
                        if (base + lane < n) wz[base + lane] = a[base + lane];
                    for (;;) {
                        const float t = 1.5f * sd;
                        const float wlo = median - t, whi = median + t;
                        int changed = 0;
                        for (int base = 0; base < n; base += 64) {
                            const int idx = base + lane;
                            const float x = idx < n ? wz[idx] : median;
                            const bool below = idx < n && x < wlo;
                            const bool above = idx < n && !below && x > whi;
                            if (below) wz[idx] = wlo;
                            if (above) wz[idx] = whi;
                            changed += __popcll(ballot64(below || above));
                        }
                        lds_fence();
                        const float ws = seq_sum(n, [&](int i) { return i < n ? wz[i] : 0.0f; });
                        const float wmean = ws / fn;
                        const float wvs = seq_sum(n, [&](int i) {
                            const float d = (i < n ? wz[i] : wmean) - wmean;
                            return i < n ? d * d : 0.0f;
                        });
                        const float old = sd;
                        sd = 1.134f * sqrt_like_go(wvs / fn);
                        const float diff = sd - old;
                        const float factor = fabsf(diff) / old;
                        if (changed == 0 || factor <= 0.0005f) break;
                    }
                }
                const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
                lo = median - t_lo;
                hi = median + t_hi;
                }
                rnd++;
                NL_T(2);

                // stack.go:411-424: swap-with-last, re-test the same index.  The loop fills every clipped position below
                // the new length m = n - c (a "hole") with a surviving sample from positions >= m, and it takes those from
                // the right end: a clipped sample that arrives in a hole is clipped again on the re-test and replaced by
                // the next one.  There are as many survivors at or behind m as holes in front of it, so the i-th hole from
                // the left receives the i-th survivor from the right -- all moves at once instead of two barriers per
                // clipped sample; the two counters count every clipped sample once either way (low tested first).
                const int before = n;
                {
                    int c = 0, c_low = 0;
                    for (int base = 0; base < n; base += 64) {
                        const int idx = base + lane;
                        const float x = idx < n ? a[idx] : 0.0f;
                        const bool low = idx < n && x < lo;
                        const bool clipped = idx < n && (low || x > hi);
                        c += __popcll(ballot64(clipped));
                        c_low += __popcll(ballot64(low));
                    }
                    c_lo += c_low;
                    c_hi += c - c_low;
                    if (c > 0) {
                        const int m = n - c;
                        int nf = 0;                                      // survivors at [m, n), listed from the left
                        for (int base = m & ~63; base < n; base += 64) {
                            const int idx = base + lane;
                            const bool in = idx >= m && idx < n;
                            const float x = in ? a[idx] : 0.0f;
                            const bool fill = in && !(x < lo || x > hi);
                            const unsigned long long mf = ballot64(fill);
                            if (fill) rfwd[nf + below64(mf)] = (unsigned short)idx;
                            nf += __popcll(mf);
                        }
                        lds_fence();
                        int nh = 0;                                      // holes at [0, m), from the left
                        for (int base = 0; base < m; base += 64) {
                            const int idx = base + lane;
                            const float x = idx < m ? a[idx] : 0.0f;
                            const bool hole = idx < m && (x < lo || x > hi);
                            const unsigned long long mh = ballot64(hole);
                            if (hole) {
                                const int src = (int)rfwd[nf - 1 - (nh + below64(mh))];
                                a[idx] = a[src];
                                if (W) wt[idx] = wt[src];
                            }
                            nh += __popcll(mh);
                        }
                        lds_fence();
                        n = m;
                    }
                }
                NL_T(3);
                if (n == before || n <= 1) {
                    res = mean;                                   // stack.go:427-430: mean before this pass
                    if constexpr (W) {                            // stack.go:514-522: weighted mean of the survivors
                        float sw, ws;
                        seq_sum2(n, [&](int i) { return i < n ? a[i] * wt[i] : 0.0f; },
                                 [&](int i) { return i < n ? wt[i] : 0.0f; }, sw, ws);
                        res = sw / ws;
                    }
                    break;
                }
            }
        }
        if (lane == 0) p.out[pix] = res;
        NL_T(4);
      }
    }
    NL_TFLUSH();
    if (lane == 0) {
        // fused pass protocol (StackArgs::final): straight to the totals; the replay of the generic pass's
        // additions is the last kernel of a pass and leaves the list lengths behind them ({exact | generic << 32}:
        // the scratch layout of nlstack_api.hip) -- read back by nl_stack_finish with the totals
        unsigned long long *slot = p.final ? p.final : p.partial + 2 * (size_t)(block % kClipSlots);
        if (c_lo) atomicAdd(slot + 0, (unsigned long long)c_lo);
        if (c_hi) atomicAdd(slot + 1, (unsigned long long)c_hi);
        if (p.final && p.list && p.list_part == 1 && block == 0)
            p.final[2] = (unsigned long long)p.list_count[0] | ((unsigned long long)p.list_count[1] << 32);
    }
}

template <bool WINSOR, bool W, int GROUP, int PF = 2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PF <= 2 ? 8 : 5, 8))) void stack_sigma_coop_kernel(StackArgs p)
{
    extern __shared__ float a[];
    coop_body<WINSOR, W, GROUP, PF>(p, a, blockIdx.x, gridDim.x);
}

#ifndef NL_TAIL_FUSED_TU          // (stack_tail_fused.hip includes this file for coop_body only)
// StackMedian (stack.go:274-303) beyond the register kernels' 512 frames: gather and the
// same wave-wide quickselect, nothing else.
__global__ __launch_bounds__(64) void stack_median_coop_kernel(StackArgs p)
{
    extern __shared__ float a[];
    unsigned short *lpos = reinterpret_cast<unsigned short *>(a + p.n_frames);
    unsigned short *rfwd = lpos + p.n_frames;
    const int lane = threadIdx.x;
    const int N = p.n_frames;
    int64_t wg = blockIdx.x;                       // XCD-contiguous pixels, see stack_sigma_coop_kernel
    if ((gridDim.x & 7u) == 0u) wg = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    for (int64_t pix = wg; pix < p.npix; pix += gridDim.x) {
        const float *fr = p.frames + pix;
        lds_fence();
        int n = 0;
        for (int base = 0; base < N; base += 64) {
            const int k = base + lane;
            const float x = k < N ? fr[(int64_t)k * p.stride] : __builtin_nanf("");
            const bool valid = x == x;
            const unsigned long long m = ballot64(valid);
            if (valid) a[n + below64(m)] = x;
            n += __popcll(m);
        }
        lds_fence();
        const float res = n > 0 ? coop_select_median(a, lpos, rfwd, n) : p.ref_loc;
        if (lane == 0) p.out[pix] = res;
    }
}

hipError_t launch_stack_median_coop(const StackArgs &args, int grid, hipStream_t stream, const char **name)
{
    *name = "stack_median_coop_kernel";
    hipLaunchKernelGGL(stack_median_coop_kernel, dim3(grid), dim3(64), (size_t)args.n_frames * 2 * sizeof(float), stream,
                       args);
    return hipGetLastError();
}

static size_t coop_columns(int mode, bool weighted)
{
    return (mode == NL_ST_WINSOR_SIGMA ? 2 : 1) + (weighted ? 1 : 0) + 1;      // samples (+copy) (+weights) + 2 scratch columns of 16 bits
}

int coop_supported(int mode, bool weighted, int n_frames)
{
    if (mode == NL_ST_MEDIAN) return (n_frames <= 65535 && (size_t)n_frames * 2 * sizeof(float) <= 64 * 1024) ? 1 : 0;
    if (mode != NL_ST_SIGMA && mode != NL_ST_WINSOR_SIGMA) return 0;
    return (n_frames <= 65535 && (size_t)n_frames * coop_columns(mode, weighted) * sizeof(float) <= 64 * 1024) ? 1 : 0;
}

// whole-tile replays take four pixels per work item when the 16-byte loads are aligned (nlstack_api.hip sizes the grid
// with the same predicate)
int coop_group(const StackArgs &args)
{
    const bool ok = args.list == nullptr && args.npix % 4 == 0 && args.stride % 4 == 0 && args.npix >= 1024 &&
                    (reinterpret_cast<uintptr_t>(args.frames) & 15u) == 0;
    return ok ? 4 : 1;
}

template <bool WINSOR, bool W>
static hipError_t launch_coop(const StackArgs &args, int grid, size_t lds, hipStream_t stream, const char **name)
{
    // (measured, profiles/r05_wsigma512_pf8_*, r05_wwinsor512_*: with PF = 8 a weighted sigma replay of 512 frames fetches
    // 1.45 x the algorithmic bytes instead of 13.7 x and takes 35.8 instead of 34.6 ms per 1024 x 4096 pixels -- it is bound by
    // the issue of the partition passes, not by its fetches; the winsorized replay, whose loops wait on memory between
    // their chains, gains 10 - 12 %: 40.5 -> 36.5 ms.  PF = 4 at 129 ... 256 frames lost 6 - 7 % in both modes.)
    static const bool deep_on = [] { const char *e = getenv("NL_COOP_PF"); return !(e && e[0] == '0'); }();      // NL_COOP_PF=0: two chunks at every depth (A/B)
    static const bool deep_all = [] { const char *e = getenv("NL_COOP_PF"); return e && e[0] == '2'; }();       // NL_COOP_PF=2: plain sigma too
    if (coop_group(args) == 4 && deep_on && (WINSOR || deep_all) && args.n_frames > 256 && args.n_frames <= 512) {
        *name = WINSOR ? (W ? "stack_sigma_coop_kernel<true, true, 4, 8>" : "stack_sigma_coop_kernel<true, false, 4, 8>")
                       : (W ? "stack_sigma_coop_kernel<false, true, 4, 8>" : "stack_sigma_coop_kernel<false, false, 4, 8>");
        hipLaunchKernelGGL((stack_sigma_coop_kernel<WINSOR, W, 4, 8>), dim3(grid), dim3(64), lds, stream, args);
    } else if (coop_group(args) == 4) {
        *name = WINSOR ? (W ? "stack_sigma_coop_kernel<true, true, 4, 2>" : "stack_sigma_coop_kernel<true, false, 4, 2>")
                       : (W ? "stack_sigma_coop_kernel<false, true, 4, 2>" : "stack_sigma_coop_kernel<false, false, 4, 2>");
        hipLaunchKernelGGL((stack_sigma_coop_kernel<WINSOR, W, 4>), dim3(grid), dim3(64), lds, stream, args);
    } else {
        *name = WINSOR ? (W ? "stack_sigma_coop_kernel<true, true, 1, 2>" : "stack_sigma_coop_kernel<true, false, 1, 2>")
                       : (W ? "stack_sigma_coop_kernel<false, true, 1, 2>" : "stack_sigma_coop_kernel<false, false, 1, 2>");
        hipLaunchKernelGGL((stack_sigma_coop_kernel<WINSOR, W, 1>), dim3(grid), dim3(64), lds, stream, args);
    }
    return hipGetLastError();
}

hipError_t launch_stack_sigma_coop(int mode, const StackArgs &args, int grid, hipStream_t stream, const char **name)
{
    const bool weighted = args.weights != nullptr;
    const size_t lds = (size_t)args.n_frames * sizeof(float) * coop_columns(mode, weighted);
    if (mode == NL_ST_WINSOR_SIGMA)
        return weighted ? launch_coop<true, true>(args, grid, lds, stream, name) : launch_coop<true, false>(args, grid, lds, stream, name);
    return weighted ? launch_coop<false, true>(args, grid, lds, stream, name) : launch_coop<false, false>(args, grid, lds, stream, name);
}

#endif  // NL_TAIL_FUSED_TU

}  // namespace nl

#if defined(NL_PROBE) && !defined(NL_TAIL_FUSED_TU)
extern "C" int nl_debug_probe(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::nl_probe_cycles), sizeof(nl::nl_probe_cycles)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nl::nl_probe_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
