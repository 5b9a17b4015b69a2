// stack_exact_coop4.hip -- bit-exact StackSigma / StackWinsorSigma replay, FOUR pixels per wavefront.
//
// stack_exact_coop.hip replays one pixel per wave; its sequential fp32 sums (stats.go:246-261: the
// reference adds in array order, so the additions cannot be re-associated) are a 63-step DPP chain per 64
// samples that keeps all 64 lanes busy for ONE useful addition per instruction -- on a winsorized 512-frame
// stack (about 20 such sums per clipping pass of a pixel) they are over half of the replay's run time
// (DESIGN.md section 5h).  Here a wave holds four pixels, one per row of 16 lanes: every phase of the replay
// (gather, Hoare partition passes, clamping, swap-with-last clipping) works on 16 samples of each pixel per
// step, and the sums are `row_shr:1` chains -- 15 steps per 16 samples for four pixels at once, 3.3 x fewer
// instructions per pixel.  Control values (left, right, k, n, ...) are uniform over a ROW and live in vector
// registers; every loop runs until no row needs it any more, rows that are done are predicated off, and the
// LDS fences sit outside all predicates.
//
// MEASURED: bit-exact (tests/test_gpu_parity.py, nl_stack_set_exact(h, 4)).  As the replay engine of the
// hand-over LISTS it is slower than one pixel per wave -- a quarter of the waves are in flight (the same LDS per
// pixel, four pixels per wave), every loop runs for the slowest of four rows, and that replay is latency-bound:
// C3 tile 5.32 -> 6.74 ms, sigma 512 tail 0.82 -> 1.45 ms (NL_COOP4=1 switches it on for A/B runs).  Over WHOLE
// tiles (weighted stacks), where the replay is bound by instruction issue, it wins from about 45 frames on:
// winsorized 96 frames 15.8 -> 9.8 ms per 512 x 4096 pixels, sigma 96 frames 5.47 -> 4.32 ms
// (tools/replay_probe2.py): dispatched there (nlstack_api.hip, stack_kernels.h).
//
// The algorithm, the visiting orders and hence every output bit and both counters are those of
// stack_exact_coop.hip (qsort.go:94-126, stats.go:246-261, stack.go:372-436, 442-531, 611-705, 710-829).
#include "stack_kernels.h"

namespace nl {

namespace {

__device__ __forceinline__ float sqrt_like_go4(float x)      // stats.go:259
{
    return (float)__builtin_sqrt((double)x);
}

__device__ __forceinline__ void fence4()
{
    __syncthreads();      // single-wave workgroup: orders LDS writes before later reads
}

// the 16 bits of a wave ballot that belong to this lane's row
__device__ __forceinline__ unsigned row_bits(unsigned long long m, int row)
{
    return (unsigned)(m >> (16 * row)) & 0xffffu;
}

// maximum of a row-uniform value over the four rows, as a wave-uniform scalar
__device__ __forceinline__ int rows_max(int v)
{
    v = max(v, __shfl_xor(v, 16, 64));
    v = max(v, __shfl_xor(v, 32, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// lane 15 of the row, in every lane of the row
__device__ __forceinline__ float row_last(float x)
{
    return __shfl(x, (int)((threadIdx.x & 48u) | 15u), 64);
}

// 16 samples per row and step: lane l of a row holds x[l]; "s[l] = s[l-1] + x[l]" issued 15 times on all lanes
// with a DPP row shift (lane 0 of every row, whose source is out of range, is left alone).  After step t the
// lanes 0..t of a row hold their final prefix sums, so lane 15 ends with the row's sequential sum.
__device__ __forceinline__ float chain16(float carry, float x, int l16)
{
    float s = (l16 == 0) ? carry + x : x;
#define NL_STEP "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
    asm volatile("s_nop 1\n\t" NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP
                 NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP NL_STEP
                 : "+v"(s) : "v"(x));
#undef NL_STEP
    return s;
}

// sequential fp32 sum of elem(0 .. n) per row (n row-uniform; elem must deliver +0.0f from n on: adding +0.0f
// leaves a running sum unchanged bit for bit, and the sum starts at +0.0f so it is never -0.0f).  The carry
// into lane 0 of a row is the previous step's lane 15 (row rotate).
template <class F>
__device__ __forceinline__ float seq_sum4(int n, int l16, F &&elem)
{
    const int top = rows_max(n);
    float s = 0.0f;
    for (int base = 0; base < top; base += 16) {
        const float carry = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x121, 0xF, 0xF, true));   // row_ror:1
        s = chain16(carry, elem(base + l16), l16);
    }
    return row_last(s);
}

// qsort.go:94-126 on a[0..n) of every row, k 1-based; see coop_select in stack_exact_coop.hip for why a
// whole Hoare partition pass can be done at once.  `on`: the row takes part.
__device__ float coop_select4(float *a, unsigned short *lpos, unsigned short *rfwd, int n, int k, bool on, int row, int l16)
{
    const unsigned below = (1u << l16) - 1u;
    int left = 0, right = n - 1;
    bool act = on && left < right;
    while (__any(act)) {
        const float pivot = act ? a[(left + right) >> 1] : 0.0f;
        int nl = 0, nr = 0;
        const int span = rows_max(act ? right - left + 1 : 0);
        for (int t = 0; t < span; t += 16) {
            const int idx = left + t + l16;
            const bool in = act && idx <= right;
            const float x = in ? a[idx] : 0.0f;
            const bool isl = in && x >= pivot;
            const bool isr = in && x <= pivot;
            const unsigned ml = row_bits(__ballot(isl), row), mr = row_bits(__ballot(isr), row);
            if (isl) lpos[nl + __popc(ml & below)] = (unsigned short)idx;
            if (isr) rfwd[nr + __popc(mr & below)] = (unsigned short)idx;          // ascending; R_i = rfwd[nr-1-i]
            nl += __popc(ml);
            nr += __popc(mr);
        }
        fence4();
        // s = number of leading pairs with L_i < R_i (the predicate is monotone in i)
        const int pairs = act ? min(nl, nr) : 0;
        int s_cnt = 0;
        const int ptop = rows_max(pairs);
        for (int t = 0; t < ptop; t += 16) {
            const int i = t + l16;
            bool ok = false;
            if (i < pairs) ok = (int)lpos[i] < (int)rfwd[nr - 1 - i];
            s_cnt += __popc(row_bits(__ballot(ok), row));
        }
        // the pass's swaps, all at once
        const int stop = rows_max(s_cnt);
        for (int t = 0; t < stop; t += 16) {
            const int i = t + l16;
            if (i < s_cnt) {
                const int pl = (int)lpos[i], pr = (int)rfwd[nr - 1 - i];
                const float xl = a[pl], xr = a[pr];
                a[pl] = xr;
                a[pr] = xl;
            }
        }
        int r = -1;
        if (act) {
            const int r_next = s_cnt < nr ? (int)rfwd[nr - 1 - s_cnt] : -1;
            const int l_prev = s_cnt > 0 ? (int)lpos[s_cnt - 1] : -1;
            r = max(r_next, l_prev);
        }
        fence4();
        if (act) {
            const int offset = r - left + 1;
            if (k <= offset) {
                right = r;
            } else {
                left = r + 1;
                k -= offset;
            }
        }
        act = act && left < right;
    }
    return on ? a[left] : 0.0f;
}

// qsort.go:68-82
__device__ float coop_select_median4(float *a, unsigned short *lpos, unsigned short *rfwd, int n, bool on, int row, int l16)
{
    const int k = (n >> 1) + 1;
    const float upper = coop_select4(a, lpos, rfwd, n, k, on, row, l16);
    // max of a[0..k-2] (even n only; computed for every row, used where n is even)
    float lower = -__builtin_inff();
    const int top = rows_max((on && !(n & 1)) ? k - 1 : 0);
    for (int t = 0; t < top; t += 16) {
        const int idx = t + l16;
        if (on && idx < k - 1) lower = fmaxf(lower, a[idx]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) lower = fmaxf(lower, __shfl_xor(lower, o, 64));
    return (n & 1) ? upper : 0.5f * (lower + upper);
}

}  // namespace

// W: weighted variants (stack.go:442-531, 710-829): the weights follow only the clip swaps (stack.go:487).
template <bool WINSOR, bool W>
__global__ __launch_bounds__(64) void stack_sigma_coop4_kernel(StackArgs p)
{
    extern __shared__ float smem[];
    constexpr int COLS = 1 + (WINSOR ? 1 : 0) + (W ? 1 : 0) + 1;      // samples (+copy) (+weights) + 2 x 16-bit partition scratch
    const int lane = threadIdx.x, row = lane >> 4, l16 = lane & 15;
    const unsigned below = (1u << l16) - 1u;
    const int N = p.n_frames;
    float *a = smem + (size_t)row * COLS * N;
    float *wz = a + N;                                         // winsorized copy (WINSOR only)
    float *wt = a + (WINSOR ? 2 : 1) * N;                      // weights (W only)
    unsigned short *lpos = reinterpret_cast<unsigned short *>(a + (COLS - 1) * N);
    unsigned short *rfwd = lpos + N;

    int64_t limit = p.npix;
    if (p.list) {
        const unsigned cnt = __atomic_load_n(p.list_count, __ATOMIC_RELAXED);
        limit = cnt < p.list_capacity ? cnt : p.list_capacity;
    }
    int64_t first = 0;
    if (p.list && p.list_snap) {                               // see stack_exact_coop.hip
        unsigned s = 0;
        if (lane == 0) {
            s = __atomic_load_n(p.list_snap, __ATOMIC_RELAXED);
            if (s == 0u) {
                s = atomicCAS(p.list_snap, 0u, (unsigned)limit + 1u);
                if (s == 0u) s = (unsigned)limit + 1u;
            }
        }
        s = (unsigned)__shfl((int)s, 0, 64);
        const int64_t snap = min((int64_t)(s - 1u), limit);
        if (p.list_part == 0) limit = snap; else first = snap;
    }
    int c_lo = 0, c_hi = 0;                                    // row-uniform

    // whole-tile replays: XCD-contiguous pixels (see stack_sigma_coop_kernel)
    int64_t wg = blockIdx.x;
    if (!p.list && (gridDim.x & 7u) == 0u) wg = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    for (int64_t item0 = first + 4 * wg; item0 < limit; item0 += 4 * (int64_t)gridDim.x) {
        const int64_t item = item0 + row;
        const bool on = item < limit;
        int64_t pix = 0;
        if (on) pix = p.list ? (int64_t)p.list[item] : item;
        const float *fr = p.frames + pix;
        fence4();
        // ---- gather in frame order, NaN dropped (stack.go:380-387) ----
        int n = 0;
        for (int base = 0; base < N; base += 16) {
            const int k = base + l16;
            const float x = (on && k < N) ? fr[(int64_t)k * p.stride] : __builtin_nanf("");
            const bool valid = x == x;
            const unsigned m = row_bits(__ballot(valid), row);
            const int pos = n + __popc(m & below);
            if (valid) a[pos] = x;
            if (W && valid) wt[pos] = p.weights[k];                          // stack.go:452-459
            n += __popc(m);
        }
        fence4();

        float res = p.ref_loc;
        bool act = on && n > 0;
        // weighted stacks with a decision pass (StackArgs::bounds): the clip bounds of a pixel's first `decided` rounds
        // are on record; in those rounds its row only permutes and clips (zero-length sums, no winsorization)
        int rnd = 0;
        int decided = 0;
        if (W && p.nrounds && on) decided = (int)p.nrounds[pix];
        while (__any(act)) {
            const bool have = W && act && rnd < decided;
            const float median = coop_select_median4(a, lpos, rfwd, n, act, row, l16);
            fence4();
            // stats.go:246-261
            const float fn = (float)n;
            const int na = (act && !have) ? n : 0;
            const float s = seq_sum4(na, l16, [&](int i) { return i < na ? a[i] : 0.0f; });
            const float mean = s / fn;
            const float vs = seq_sum4(na, l16, [&](int i) {
                const float d = (i < na ? a[i] : mean) - mean;
                return i < na ? d * d : 0.0f;
            });
            const float var = vs / fn;
            float sd = sqrt_like_go4(var);
            if constexpr (WINSOR) {
                // stack.go:646-672
                const int top = rows_max(na);
                for (int t = 0; t < top; t += 16)
                    if (t + l16 < na) wz[t + l16] = a[t + l16];
                bool inner = act && !have;
                while (__any(inner)) {
                    const float tt = 1.5f * sd;
                    const float wlo = median - tt, whi = median + tt;
                    int changed = 0;
                    const int ni = inner ? n : 0;
                    const int itop = rows_max(ni);
                    for (int t = 0; t < itop; t += 16) {
                        const int idx = t + l16;
                        const float x = idx < ni ? wz[idx] : median;
                        const bool lowc = idx < ni && x < wlo;
                        const bool highc = idx < ni && !lowc && x > whi;
                        if (lowc) wz[idx] = wlo;
                        if (highc) wz[idx] = whi;
                        changed += __popc(row_bits(__ballot(lowc || highc), row));
                    }
                    fence4();
                    const float ws = seq_sum4(ni, l16, [&](int i) { return i < ni ? wz[i] : 0.0f; });
                    const float wmean = ws / fn;
                    const float wvs = seq_sum4(ni, l16, [&](int i) {
                        const float d = (i < ni ? wz[i] : wmean) - wmean;
                        return i < ni ? d * d : 0.0f;
                    });
                    if (inner) {
                        const float old = sd;
                        sd = 1.134f * sqrt_like_go4(wvs / fn);
                        const float diff = sd - old;
                        const float factor = fabsf(diff) / old;
                        if (changed == 0 || factor <= 0.0005f) inner = false;
                    }
                }
            }
            const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
            float lo = median - t_lo, hi = median + t_hi;
            if (have) {
                const float2 bd = p.bounds[(size_t)rnd * (size_t)p.npix + (size_t)pix];
                lo = bd.x;
                hi = bd.y;
            }
            rnd++;

            // stack.go:411-424: swap-with-last, re-test the same index.  Every trip of the loop a row either
            // scans one stretch of 16 for its next reject or removes the one it found.
            const int before = n;
            int base = 0;
            bool scanning = act;
            while (__any(scanning)) {
                const int idx = base + l16;
                const float x = (scanning && idx < n) ? a[idx] : 0.0f;
                const bool clipped = scanning && idx < n && (x < lo || x > hi);
                const unsigned m = row_bits(__ballot(clipped), row);
                const bool hit = scanning && m != 0u;
                int found = 0;
                float last = 0.0f, last_w = 0.0f;
                if (hit) {
                    found = base + (int)__builtin_ctz(m);
                    const float g = a[found];
                    last = a[n - 1];
                    if (W) last_w = wt[n - 1];
                    if (g < lo) c_lo++; else c_hi++;
                }
                fence4();
                if (hit) {
                    if (l16 == 0) { a[found] = last; if (W) wt[found] = last_w; }
                    n--;
                    base = found;
                } else if (scanning) {
                    base += 16;
                }
                fence4();
                scanning = scanning && base < n;
            }
            const bool fin = act && (n == before || n <= 1);
            if (__any(fin)) {
                if constexpr (W) {                             // stack.go:514-522: weighted mean of the survivors
                    const int nf = fin ? n : 0;
                    const float sw = seq_sum4(nf, l16, [&](int i) { return i < nf ? a[i] * wt[i] : 0.0f; });
                    const float ws = seq_sum4(nf, l16, [&](int i) { return i < nf ? wt[i] : 0.0f; });
                    if (fin) res = sw / ws;
                } else {
                    if (fin) res = mean;                       // stack.go:427-430: mean before this pass
                }
                act = act && !fin;
            }
        }
        if (on && l16 == 0) p.out[pix] = res;
    }
    if (l16 == 0) {
        unsigned long long *slot = p.final ? p.final : p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (c_lo) atomicAdd(slot + 0, (unsigned long long)c_lo);
        if (c_hi) atomicAdd(slot + 1, (unsigned long long)c_hi);
        if (p.final && p.list && p.list_part == 1 && blockIdx.x == 0 && lane == 0)
            p.final[2] = (unsigned long long)p.list_count[0] | ((unsigned long long)p.list_count[1] << 32);
    }
}

static size_t coop4_columns(int mode, bool weighted)
{
    return (mode == NL_ST_WINSOR_SIGMA ? 2 : 1) + (weighted ? 1 : 0) + 1;
}

int coop4_supported(int mode, bool weighted, int n_frames)
{
    if (mode != NL_ST_SIGMA && mode != NL_ST_WINSOR_SIGMA) return 0;
    return (n_frames <= 65535 && 4 * (size_t)n_frames * coop4_columns(mode, weighted) * sizeof(float) <= 64 * 1024) ? 1 : 0;
}

hipError_t launch_stack_sigma_coop4(int mode, const StackArgs &args, int grid, hipStream_t stream, const char **name)
{
    const bool weighted = args.weights != nullptr;
    const size_t lds = 4 * (size_t)args.n_frames * sizeof(float) * coop4_columns(mode, weighted);
    if (mode == NL_ST_WINSOR_SIGMA) {
        if (weighted) {
            *name = "stack_sigma_coop4_kernel<true, true>";
            hipLaunchKernelGGL((stack_sigma_coop4_kernel<true, true>), dim3(grid), dim3(64), lds, stream, args);
        } else {
            *name = "stack_sigma_coop4_kernel<true, false>";
            hipLaunchKernelGGL((stack_sigma_coop4_kernel<true, false>), dim3(grid), dim3(64), lds, stream, args);
        }
    } else {
        if (weighted) {
            *name = "stack_sigma_coop4_kernel<false, true>";
            hipLaunchKernelGGL((stack_sigma_coop4_kernel<false, true>), dim3(grid), dim3(64), lds, stream, args);
        } else {
            *name = "stack_sigma_coop4_kernel<false, false>";
            hipLaunchKernelGGL((stack_sigma_coop4_kernel<false, false>), dim3(grid), dim3(64), lds, stream, args);
        }
    }
    return hipGetLastError();
}

}  // namespace nl
