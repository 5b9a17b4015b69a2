// stack_exact_tile.hip -- bit-exact StackSigma / StackWinsorSigma (+Weighted) over whole tiles.
//
// The weighted clip modes cannot use the register-resident kernels: the reference permutes the
// samples with quickselect but NOT the weights (internal/ops/stack/stack.go:487 -- the weights
// only follow the clip swaps), and the final weighted mean pairs sample i of the permuted array
// with weight i all the same (stack.go:514-522).  The result therefore depends on the exact
// permutation Hoare's partition leaves, and has to be replayed.  So does a stack beyond the
// register kernels' 512 frames.
//
// Layout.  One wavefront owns 64 CONSECUTIVE pixels: every frame contributes one coalesced
// 256-byte row segment, read from HBM exactly once (4*(N+1) B per pixel, the algorithmic
// traffic).  The samples go to LDS as [slot][lane]: a lane's column is its pixel, bank = lane
// for any per-lane slot index, so the data-dependent accesses of quickselect and of
// swap-with-last never conflict.  One pixel per lane from there on:
//   gather       stack.go:380-387      frame order, NaN dropped (predicated store, per-lane cursor)
//   quickselect  qsort.go:94-126       a per-lane state machine: every step reads two candidates
//                                      for each scan pointer (and the next pivot) and all lanes
//                                      advance -- no nested data-dependent loops, so 64 different
//                                      permutations replay in lock step
//   mean/stddev  stats.go:246-261      sequential fp32 sums in index order; the loop bounds are
//                                      wave-uniform, a lane past its n adds +0.0f (bitwise neutral)
//   winsorize    stack.go:646-672      the running clamp (stack_exact.hip:winsorized_stddev)
//   clip         stack.go:411-424      the two counters are order independent and come from a
//                                      plain scan; only the arrangement swap-with-last leaves is
//                                      replayed, four positions per step while nothing is clipped
//   weighted mean stack.go:514-522     weights as frame indices in a byte column that follows
//                                      the clip swaps
// Same fp32 operations in the same order as the reference (no FMA contraction): every output
// bit and both counters equal the oracle's.
#include "stack_kernels.h"

namespace nl {

namespace {

constexpr int TS = 64;     // lanes = pixels per tile = LDS row length in floats

__device__ __forceinline__ float sqrt_go32(float x)      // float32(math.Sqrt(float64(x))), stats.go:259
{
    return (float)__builtin_sqrt((double)x);
}

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// row index clamped into the column: speculative reads (a second scan candidate, a lane that
// has nothing to do this step) must stay inside the allocation; what they return is never used
__device__ __forceinline__ int row(int i, int top) { return min(max(i, 0), top) * TS; }

// qsort.go:94-126 for 64 columns at once; a = the lane's column (rows 0 .. top), k 1-based,
// act = lane takes part.  Returns a[left] of the lane's final range.
__device__ float tile_select(float *a, int n, int k, bool act, int top)
{
    int left = 0, right = act ? n - 1 : 0;
    bool busy = act && left < right;
    int l = left - 1, r = right + 1;
    float pivot = 0.0f, al = 0.0f, ar = 0.0f;
    bool need_pivot = true, lstop = false, rstop = false;
    while (__any(busy)) {
        // every lane reads; what a lane has no use for is ignored
        const float pv = a[row((left + right) >> 1, top)];
        const float x1 = a[row(l + 1, top)], x2 = a[row(l + 2, top)];
        const float y1 = a[row(r - 1, top)], y2 = a[row(r - 2, top)];
        if (need_pivot) { pivot = pv; need_pivot = false; }
        if (busy) {
            // do { l++; al = a[l]; } while (!(al >= pivot) && l < right);   two candidates per step
            if (!lstop) {
                const bool s1 = (x1 >= pivot) || (l + 1 >= right);
                const bool s2 = (x2 >= pivot) || (l + 2 >= right);
                if (s1) { l += 1; al = x1; lstop = true; }
                else { l += 2; al = x2; lstop = s2; }
            }
            // do { r--; ar = a[r]; } while (!(ar <= pivot) && r > left);
            if (!rstop) {
                const bool s1 = (y1 <= pivot) || (r - 1 <= left);
                const bool s2 = (y2 <= pivot) || (r - 2 <= left);
                if (s1) { r -= 1; ar = y1; rstop = true; }
                else { r -= 2; ar = y2; rstop = s2; }
            }
            if (lstop && rstop) {
                if (l >= r) {                       // the pass is over: qsort.go:115-123
                    const int offset = r - left + 1;
                    if (k <= offset) right = r;
                    else { left = r + 1; k -= offset; }
                    busy = left < right;
                    l = left - 1; r = right + 1;
                    need_pivot = true;
                } else {                            // swap and go on scanning
                    a[l * TS] = ar;
                    a[r * TS] = al;
                }
                lstop = rstop = false;
            }
        }
    }
    return a[row(left, top)];
}

// sequential fp32 sum over a lane's first n rows of f(row value, row index): uniform bound
// nmax, 8 rows per trip so that the loads of a trip are in flight together
template <class F>
__device__ __forceinline__ float seq_rows(const float *a, int n, int nmax, int top, F &&term)
{
    float s = 0.0f;
    for (int base = 0; base < nmax; base += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = a[min(base + u, top) * TS];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float t = term(x[u], base + u);
            s += (base + u < n) ? t : 0.0f;
        }
    }
    return s;
}

}  // namespace

// IDX: type of the frame-index column (weighted only): unsigned char up to 256 frames
template <bool WINSOR, bool W, class IDX>
__global__ __launch_bounds__(64) void stack_sigma_tile_kernel(StackArgs p)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int N = p.n_frames;
    const int top = N - 1;                                   // rows 0 .. N-1, nothing to spare (N = 128 weighted:
    float *a = lds + lane;                                   // 40 KiB per tile = four tiles per CU)
    IDX *ix = reinterpret_cast<IDX *>(lds + (size_t)N * TS) + lane;         // ix[i * TS]

    int64_t limit = p.npix;
    if (p.list) {
        const unsigned cnt = *p.list_count;
        limit = cnt < p.list_capacity ? cnt : p.list_capacity;
    }
    const int64_t tiles = (limit + TS - 1) / TS;
    int c_lo = 0, c_hi = 0;

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t item = tile * TS + lane;
        const bool on = item < limit;
        const int64_t pix = p.list ? (int64_t)p.list[on ? item : 0] : (on ? item : 0);
        const float *fr = p.frames + pix;

        // ---- gather (stack.go:380-387 / 452-459): frame order, NaN dropped ----
        int n = 0;
        int k = 0;
        for (; k + 8 <= N; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(fr + (int64_t)(k + u) * p.stride);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (v[u] == v[u]) {
                    a[n * TS] = v[u];
                    if (W) ix[n * TS] = (IDX)(k + u);
                    n++;
                }
            }
        }
        for (; k < N; k++) {
            const float v = __builtin_nontemporal_load(fr + (int64_t)k * p.stride);
            if (v == v) {
                a[n * TS] = v;
                if (W) ix[n * TS] = (IDX)k;
                n++;
            }
        }

        float res = p.ref_loc;                               // stack.go:388-397
        bool act = on && n > 0;
        while (__any(act)) {
            const int nmax = wave_max_i(act ? n : 0);
            const float fn = (float)n;
            // ---- median (qsort.go:68-82) ----
            const int kk = (n >> 1) + 1;
            const float upper = tile_select(a, n, kk, act, top);
            float lower = -__builtin_inff();
            {
                const int km = wave_max_i((act && !(n & 1)) ? kk - 1 : 0);
                for (int base = 0; base < km; base += 8) {
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = a[min(base + u, top) * TS];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (base + u < kk - 1 && x[u] > lower) lower = x[u];
                }
            }
            const float median = (n & 1) ? upper : 0.5f * (lower + upper);
            // ---- MeanStdDev (stats.go:246-261) ----
            const float s = seq_rows(a, n, nmax, top, [](float x, int) { return x; });
            const float mean = s / fn;
            const float vs = seq_rows(a, n, nmax, top, [mean](float x, int) { const float d = x - mean; return d * d; });
            float sd = sqrt_go32(vs / fn);
            if constexpr (WINSOR) {
                // stack.go:646-672 with the running clamp of stack_exact.hip:winsorized_stddev
                float Leff = -__builtin_inff(), Heff = __builtin_inff();
                bool wact = act;
                while (__any(wact)) {
                    const float t = 1.5f * sd;
                    const float lo = median - t, hi = median + t;
                    const float Lnew = fmaxf(Leff, lo), Hnew = fminf(Heff, hi);
                    const bool ok = (lo == lo);
                    int changed = 0;
                    const float ws = seq_rows(a, n, nmax, top, [&](float x, int i) {
                        float wz = fminf(fmaxf(x, Leff), Heff);
                        if (ok) {
                            if (wz < lo) { wz = lo; changed += (i < n) ? 1 : 0; }
                            else if (wz > hi) { wz = hi; changed += (i < n) ? 1 : 0; }
                        }
                        return wz;
                    });
                    const float Lc = ok ? Lnew : Leff, Hc = ok ? Hnew : Heff;
                    const float wm = ws / fn;
                    const float wv = seq_rows(a, n, nmax, top, [&](float x, int) {
                        const float wz = fminf(fmaxf(x, Lc), Hc);
                        const float d = wz - wm;
                        return d * d;
                    });
                    const float sdn = 1.134f * sqrt_go32(wv / fn);
                    const float factor = fabsf(sdn - sd) / sd;
                    if (wact) {
                        Leff = Lc; Heff = Hc;
                        sd = sdn;
                        if (changed == 0 || factor <= 0.0005f) wact = false;
                    }
                }
            }
            const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
            const float lo = median - t_lo, hi = median + t_hi;

            // ---- the two counters (stack.go:413-422): low is tested first; order independent ----
            int clo = 0, chi = 0;
            for (int base = 0; base < nmax; base += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = a[min(base + u, top) * TS];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const bool in = base + u < n;
                    const bool low = in && x[u] < lo;
                    const bool high = in && !low && x[u] > hi;
                    clo += low ? 1 : 0;
                    chi += high ? 1 : 0;
                }
            }
            const int clipped = act ? clo + chi : 0;
            // ---- the arrangement swap-with-last leaves (stack.go:411-424) ----
            // while j < len: if a[j] is clipped { a[j] = a[len-1]; len-- (re-test j) } else j++.
            // Four positions per step while nothing is clipped; a[len-1] is read with them, so
            // the replacement happens in the step that finds the clipped sample.
            {
                int j = 0, m = n, found = 0;
                bool cb = clipped > 0;
                while (__any(cb)) {
                    const float g0 = a[row(j, top)], g1 = a[row(j + 1, top)];
                    const float g2 = a[row(j + 2, top)], g3 = a[row(j + 3, top)];
                    const float last = a[row(m - 1, top)];
                    IDX last_ix = 0;
                    if (W) last_ix = ix[row(m - 1, top)];
                    if (cb) {
                        const bool q0 = (g0 < lo || g0 > hi);                  // j < m holds while cb
                        const bool q1 = j + 1 < m && (g1 < lo || g1 > hi);
                        const bool q2 = j + 2 < m && (g2 < lo || g2 > hi);
                        const bool q3 = j + 3 < m && (g3 < lo || g3 > hi);
                        int adv = 4;
                        if (q3) adv = 3;
                        if (q2) adv = 2;
                        if (q1) adv = 1;
                        if (q0) adv = 0;
                        j += adv;
                        if (adv < 4) {
                            a[j * TS] = last;
                            if (W) ix[j * TS] = last_ix;
                            m--;
                            found++;
                        }
                        if (j >= m || found == clipped) cb = false;
                    }
                }
            }
            if (act) {
                c_lo += clo;
                c_hi += chi;
            }
            const int n_new = n - clipped;
            const bool done = act && (clipped == 0 || n_new <= 1);           // stack.go:427
            if constexpr (W) {
                // stack.go:514-522: weighted mean of the survivors, index by index
                if (__any(done)) {
                    const int wmax = wave_max_i(done ? n_new : 0);
                    float sw = 0.0f, wsum = 0.0f;
                    for (int base = 0; base < wmax; base += 8) {
                        float x[8], w[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            x[u] = a[min(base + u, top) * TS];
                            const int f = (int)ix[min(base + u, top) * TS];
                            w[u] = p.weights[min(f, N - 1)];
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const bool in = base + u < n_new;
                            const float pr = x[u] * w[u];
                            sw += in ? pr : 0.0f;
                            wsum += in ? w[u] : 0.0f;
                        }
                    }
                    if (done) res = sw / wsum;
                }
            } else {
                if (done) res = mean;                         // stack.go:427-430: the mean BEFORE this pass
            }
            if (done) act = false;
            if (act) n = n_new;
        }
        if (on) p.out[pix] = res;
    }

    const int t_lo = wave_sum_i(c_lo), t_hi = wave_sum_i(c_hi);
    if (lane == 0) {
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}

// LDS bytes of one tile, 0 if it does not fit the CU
size_t tile_lds_bytes(int n_frames, bool weighted)
{
    const size_t rows = (size_t)n_frames;
    size_t bytes = rows * TS * sizeof(float);
    if (weighted) bytes += rows * TS * (n_frames <= 256 ? 1 : 2);
    return bytes <= kLdsBudgetBytes ? bytes : 0;
}

int tile_supported(int mode, bool weighted, int n_frames)
{
    if (mode != NL_ST_SIGMA && mode != NL_ST_WINSOR_SIGMA) return 0;
    if (n_frames > 65535) return 0;
    return tile_lds_bytes(n_frames, weighted) != 0 ? 1 : 0;
}

template <bool WINSOR, bool W, class IDX>
static hipError_t launch_tile(const StackArgs &args, int grid, size_t lds, hipStream_t stream)
{
    auto kern = stack_sigma_tile_kernel<WINSOR, W, IDX>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, args);
    return hipGetLastError();
}

hipError_t launch_stack_sigma_tile(int mode, const StackArgs &args, int grid, hipStream_t stream, const char **name)
{
    const bool weighted = args.weights != nullptr;
    const size_t lds = tile_lds_bytes(args.n_frames, weighted);
    if (!lds) return hipErrorInvalidValue;
    const bool winsor = mode == NL_ST_WINSOR_SIGMA;
    if (!weighted) {
        *name = winsor ? "stack_sigma_tile_kernel<winsor>" : "stack_sigma_tile_kernel<sigma>";
        return winsor ? launch_tile<true, false, unsigned char>(args, grid, lds, stream)
                      : launch_tile<false, false, unsigned char>(args, grid, lds, stream);
    }
    *name = winsor ? "stack_sigma_tile_kernel<winsor,weighted>" : "stack_sigma_tile_kernel<sigma,weighted>";
    if (args.n_frames <= 256)
        return winsor ? launch_tile<true, true, unsigned char>(args, grid, lds, stream)
                      : launch_tile<false, true, unsigned char>(args, grid, lds, stream);
    return winsor ? launch_tile<true, true, unsigned short>(args, grid, lds, stream)
                  : launch_tile<false, true, unsigned short>(args, grid, lds, stream);
}

}  // namespace nl
