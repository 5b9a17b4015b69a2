// stack_exact.hip -- bit-exact per-pixel stacking kernels for gfx950 (MI355X).
//
// One wavefront per workgroup, one pixel per lane.  Each lane owns a column of
// the workgroup's LDS tile laid out [slot][lane] (bank = lane, so the
// data-dependent slot index of quickselect / swap-with-last never conflicts),
// and runs the reference's per-pixel algorithm in the reference's order, so
// that fp32 sums see the same permutation and results match bit for bit:
//   gather        internal/ops/stack/stack.go:380-387 (frame order, NaN skipped)
//   quickselect   internal/qsort/qsort.go:68-126     (Hoare, middle pivot, in place)
//   mean/stddev   internal/stats/stats.go:246-261    (two sequential fp32 passes)
//   sigma clip    stack.go:401-431, weighted :484-524
//   winsorized    stack.go:641-698, weighted :753-821
//   MAD           stack.go:565-600
//   linear fit    stack.go:869-911, stats.go:569-586
//   median        stack.go:274-303
// Build flags that matter for parity: -ffp-contract=off (Go/amd64 never fuses
// a*b+c) and -fhip-fp32-correctly-rounded-divide-sqrt (IEEE divide / sqrt).
//
// Roofline: HBM-bound by construction (4*(N+1) bytes per output pixel, each
// sample read exactly once with 256-byte coalesced rows per frame); in this
// exact form the kernel is limited by per-lane LDS latency and VALU issue,
// see DESIGN.md section 4.
#include "stack_kernels.h"

namespace nl {

// ---- a lane's column in the [slot][lane] LDS tile -------------------------
template <int S>
struct Col {
    float *p;
    bool live;      // lanes beyond LANES (narrow tiles) alias lane 0's column: they may read it, never write it
    __device__ __forceinline__ float get(int i) const { return p[i * S]; }
    __device__ __forceinline__ void set(int i, float v) const { if (live) p[i * S] = v; }
    __device__ __forceinline__ Col<S> shifted(int k) const { return Col<S>{p + k * S, live}; }
};

// qsort.go:94-126, k is 1-based
template <int S>
__device__ float hoare_select(Col<S> a, int n, int k)
{
    int left = 0, right = n - 1;
    while (left < right) {
        const float pivot = a.get((left + right) >> 1);
        int l = left - 1, r = right + 1;
        for (;;) {
            float al, ar;
            // (the index guards never fire on NaN-free data; they keep a NaN -- only possible in
            // MAD deviations of a pixel with an infinite median, where the reference itself dies
            // with an index panic -- from sending the scan past the column forever)
            do { l++; al = a.get(l); } while (!(al >= pivot) && l < right);
            do { r--; ar = a.get(r); } while (!(ar <= pivot) && r > left);
            if (l >= r) break;
            a.set(l, ar);
            a.set(r, al);
        }
        const int offset = r - left + 1;
        if (k <= offset) {
            right = r;
        } else {
            left = r + 1;
            k -= offset;
        }
    }
    return a.get(left);
}

// qsort.go:68-82
template <int S>
__device__ float select_median(Col<S> a, int n)
{
    const int k = (n >> 1) + 1;
    const float upper = hoare_select(a, n, k);
    float res = upper;
    if ((n & 1) == 0) {
        float lower = a.get(0);
        for (int i = 1; i < k - 1; i++) {
            const float v = a.get(i);
            if (v > lower) lower = v;
        }
        res = 0.5f * (lower + upper);
    }
    return res;
}

// float32(math.Sqrt(float64(x))) (stats.go:259): square root in fp64, rounded
// once to fp32 = the correctly rounded fp32 root.  NOT __fsqrt_rn: without
// OCML_BASIC_ROUNDED_OPERATIONS HIP maps that to the 1-ulp native v_sqrt_f32,
// which flipped one clip decision in 2.1e9 samples against the oracle.
__device__ __forceinline__ float sqrt_like_go(float x)
{
    return (float)__builtin_sqrt((double)x);
}

// stats.go:246-261
template <int S>
__device__ void mean_stddev(Col<S> a, int n, float &mean, float &sd)
{
    float s = 0.0f;
    for (int i = 0; i < n; i++) s += a.get(i);
    const float fn = (float)n;
    const float m = s / fn;
    float v = 0.0f;
    for (int i = 0; i < n; i++) {
        const float d = a.get(i) - m;
        v += d * d;
    }
    v = v / fn;
    mean = m;
    sd = sqrt_like_go(v);
}

// stack.go:411-424 (+ :494-511 with the mirrored weight column)
template <int S, bool W>
__device__ int clip_pass(Col<S> a, Col<S> w, int n, float lo, float hi, int &c_lo, int &c_hi)
{
    int j = 0;
    while (j < n) {
        const float g = a.get(j);
        const bool low = g < lo;
        const bool high = !low && (g > hi);
        if (low || high) {
            n--;
            a.set(j, a.get(n));
            if (W) w.set(j, w.get(n));
            c_lo += low ? 1 : 0;
            c_hi += high ? 1 : 0;
        } else {
            j++;
        }
    }
    return n;
}

// stack.go:514-522
template <int S>
__device__ float weighted_mean(Col<S> a, Col<S> w, int n)
{
    float s = 0.0f, ws = 0.0f;
    for (int i = 0; i < n; i++) {
        const float wi = w.get(i);
        const float p = a.get(i) * wi;
        s += p;
        ws += wi;
    }
    return s / ws;
}

// stack.go:646-672.  The reference copies the column and clamps the copy
// repeatedly; all clamp intervals share the centre `median`, so after any
// number of rounds the copy equals clamp(a[i], max(lo_j), min(hi_j)).  We keep
// only those two running bounds and re-derive the copy on the fly: identical
// values in identical order, no second column.
template <int S>
__device__ float winsorized_stddev(Col<S> a, int n, float median, float sd)
{
    float Leff = -__builtin_inff(), Heff = __builtin_inff();
    const float fn = (float)n;
    for (;;) {
        const float t = 1.5f * sd;
        const float lo = median - t, hi = median + t;
        int changed = 0;
        float s = 0.0f;
        const float Lnew = fmaxf(Leff, lo), Hnew = fminf(Heff, hi);
        const bool ok = (lo == lo);   // NaN stddev: every comparison false, nothing changes
        for (int i = 0; i < n; i++) {
            const float x = a.get(i);
            float wz = fminf(fmaxf(x, Leff), Heff);       // copy before this round
            if (ok) {
                if (wz < lo) { wz = lo; changed++; }
                else if (wz > hi) { wz = hi; changed++; }
            }
            s += wz;
        }
        if (ok) { Leff = Lnew; Heff = Hnew; }
        const float m = s / fn;
        float v = 0.0f;
        for (int i = 0; i < n; i++) {
            const float wz = fminf(fmaxf(a.get(i), Leff), Heff);
            const float d = wz - m;
            v += d * d;
        }
        v = v / fn;
        const float old = sd;
        sd = 1.134f * sqrt_like_go(v);
        const float factor = fabsf(sd - old) / old;
        if (changed == 0 || factor <= 0.0005f) break;
    }
    return sd;
}

// In-column bitonic sort, ascending, n_pad = power of two; every lane runs the
// same compare-exchange sequence (no divergence, no bank conflicts).  A sorted
// column is unique, so any sorting algorithm reproduces qsort.go:26-32 exactly.
template <int S>
__device__ void bitonic_sort(Col<S> a, int n_pad)
{
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int half = n_pad >> 1;
            for (int t = 0; t < half; t++) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const float x = a.get(i), y = a.get(l);
                const float mn = fminf(x, y), mx = fmaxf(x, y);
                const bool asc = (i & k) == 0;
                a.set(i, asc ? mn : mx);
                a.set(l, asc ? mx : mn);
            }
        }
    }
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- the kernel -----------------------------------------------------------
// MODE: NL_ST_* (0 median, 2 sigma, 3 winsor, 4 MAD, 5 linear fit);
// W: weighted (sigma / winsor only); LANES: pixels per wavefront (64/32/16,
// smaller when a 64-wide tile would not fit the 160 KiB LDS).
template <int MODE, bool W, int LANES>
__global__ __launch_bounds__(64) void stack_exact_kernel(StackArgs p)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const bool lane_on = lane < LANES;
    const int n_alloc = (MODE == NL_ST_LINEAR_FIT) ? p.n_pad : p.n_frames;
    Col<LANES> a{lds + (lane_on ? lane : 0), lane_on};
    Col<LANES> b{lds + (size_t)n_alloc * LANES + (lane_on ? lane : 0), lane_on};   // weights / abs-dev column

    int c_lo = 0, c_hi = 0;

    // list mode: redo only the pixels a fast kernel could not decide
    int64_t limit = p.npix, tiles = p.tiles;
    if (p.list) {
        const unsigned cnt = *p.list_count;
        limit = cnt < p.list_capacity ? cnt : p.list_capacity;
        tiles = (limit + LANES - 1) / LANES;
    }

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t idx = tile * LANES + lane;
        const bool on = lane_on && idx < limit;
        const int64_t pix = p.list ? (int64_t)p.list[on ? idx : 0] : idx;
        const float *fr = p.frames + (on ? pix : 0);

        // ---- gather (stack.go:380-387): frame order, NaN dropped ----
        int n = 0;
        int k = 0;
        const int N = p.n_frames;
        for (; k + 8 <= N; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = fr[(int64_t)(k + u) * p.stride];
            if (on) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (v[u] == v[u]) {
                        a.set(n, v[u]);
                        if (W) b.set(n, p.weights[k + u]);
                        n++;
                    }
                }
            }
        }
        for (; k < N; k++) {
            const float v = fr[(int64_t)k * p.stride];
            if (on && v == v) {
                a.set(n, v);
                if (W) b.set(n, p.weights[k]);
                n++;
            }
        }

        float res = p.ref_loc;     // stack.go:388-397: no valid sample -> RefFrameLoc

        if (MODE == NL_ST_MEDIAN) {
            if (n > 0) res = select_median(a, n);
        } else if (MODE == NL_ST_SIGMA || MODE == NL_ST_WINSOR_SIGMA) {
            if (n > 0) {
                for (;;) {
                    const float median = select_median(a, n);
                    float mean, sd;
                    mean_stddev(a, n, mean, sd);
                    if (MODE == NL_ST_WINSOR_SIGMA) sd = winsorized_stddev(a, n, median, sd);
                    const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
                    const float lo = median - t_lo, hi = median + t_hi;
                    const int before = n;
                    n = clip_pass<LANES, W>(a, b, n, lo, hi, c_lo, c_hi);
                    if (n == before || n <= 1) {
                        res = W ? weighted_mean(a, b, n) : mean;
                        break;
                    }
                }
            }
        } else if (MODE == NL_ST_MAD_SIGMA) {
            if (n > 0) {
                const float median = select_median(a, n);
                for (int i = 0; i < n; i++) {
                    float d = a.get(i) - median;
                    if (d < 0) d = -d;
                    b.set(i, d);
                }
                const float mad = select_median(b, n);
                const float sd = mad * 1.4826f;
                const float t_lo = p.sig_lo * sd, t_hi = p.sig_hi * sd;
                const float lo = median - t_lo, hi = median + t_hi;
                n = clip_pass<LANES, false>(a, b, n, lo, hi, c_lo, c_hi);
                float s = 0.0f;
                for (int i = 0; i < n; i++) s += a.get(i);
                res = s / (float)n;
            }
        } else if (MODE == NL_ST_LINEAR_FIT) {
            // pad the column with +Inf up to n_pad and sort once; rejects are
            // removed by stable compaction, which keeps the column sorted, so
            // the reference's per-iteration re-sort (stack.go:872) is implied.
            for (int i = n; i < p.n_pad; i++) a.set(i, __builtin_inff());
            bitonic_sort(a, p.n_pad);
            if (n > 0) {
                float mean = 0.0f;
                for (;;) {
                    const float fn = (float)n;
                    // stats.go:570 on xs = 0..n-1: depends on n only, tabulated on the host
                    const float xm = p.xstat[2 * n], xsd = p.xstat[2 * n + 1];
                    float ym, ysd;
                    mean_stddev(a, n, ym, ysd);
                    float corr = 0.0f;
                    for (int i = 0; i < n; i++) {
                        const float dx = (float)i - xm;
                        const float dy = a.get(i) - ym;
                        const float d = dx * dy;
                        corr += d;
                    }
                    float den = xsd * ysd;
                    den = den * (fn + 1.0f);
                    corr = corr / den;
                    float slope = corr * ysd;
                    slope = slope / xsd;
                    const float sx = slope * xm;
                    const float icpt = ym - sx;
                    mean = ym;
                    float sg = 0.0f;
                    for (int i = 0; i < n; i++) {
                        float lin = (float)i * slope;
                        lin = lin + icpt;
                        const float diff = a.get(i) - lin;
                        sg += fabsf(diff);
                    }
                    sg = sg / fn;
                    const float lb = p.sig_lo * sg, hb = p.sig_hi * sg;
                    int kept = 0;
                    for (int i = 0; i < n; i++) {
                        const float g = a.get(i);
                        float lin = (float)i * slope;
                        lin = lin + icpt;
                        const bool low = (lin - g) > lb;
                        const bool high = !low && ((g - lin) > hb);
                        if (low) c_lo++;
                        else if (high) c_hi++;
                        else { a.set(kept, g); kept++; }
                    }
                    const int left = n - kept;
                    if (left == 0 || n < 3) break;
                    n = kept;
                }
                res = mean;
            }
        }

        if (on) p.out[pix] = res;
    }

    // clip totals (stack.go:193-198): wave sum -> one integer atomic per
    // workgroup into kClipSlots sharded accumulators (integer adds commute:
    // the totals are deterministic), summed by reduce_counters_kernel
    const int t_lo = wave_sum(c_lo), t_hi = wave_sum(c_hi);
    if (lane == 0) {
        unsigned long long *slot = p.partial + 2 * (size_t)(blockIdx.x % kClipSlots);
        if (t_lo) atomicAdd(slot + 0, (unsigned long long)t_lo);
        if (t_hi) atomicAdd(slot + 1, (unsigned long long)t_hi);
    }
}

// zero_after: the scratch set goes back to all zeros behind the sums -- the clip slots and the two words of list
// lengths / snapshot behind them (kScratchWords) -- so that the next pass on the handle needs no memset in front
__global__ __launch_bounds__(256) void reduce_counters_kernel(unsigned long long *partial,
                                                               int n_slots,
                                                               unsigned long long *counters,
                                                               const unsigned *list_counts, int n_lists, int list_stride,
                                                               int zero_after)
{
    __shared__ unsigned long long s_lo[256], s_hi[256];
    unsigned long long lo = 0, hi = 0;
    for (int i = threadIdx.x; i < n_slots; i += 256) {
        lo += partial[2 * (size_t)i];
        hi += partial[2 * (size_t)i + 1];
    }
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    if (zero_after)
        for (int i = threadIdx.x; i < 2 * n_slots; i += 256) partial[i] = 0ull;
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_lo[threadIdx.x] += s_lo[threadIdx.x + off];
            s_hi[threadIdx.x] += s_hi[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counters[0] = s_lo[0];
        counters[1] = s_hi[0];
        if (list_counts) {             // (chunked passes keep one pair of list lengths per chunk)
            unsigned fb = 0, gen = 0;
            for (int i = 0; i < n_lists; i++) { fb += list_counts[i * list_stride]; gen += list_counts[i * list_stride + 1]; }
            counters[2] = (unsigned long long)fb | ((unsigned long long)gen << 32);
        }
        if (zero_after) { partial[2 * n_slots] = 0ull; partial[2 * n_slots + 1] = 0ull; }
    }
}

// ---- host-side launcher ----------------------------------------------------
template <int MODE, bool W, int LANES>
static hipError_t launch_exact(const StackArgs &args, int grid, size_t lds_bytes, hipStream_t stream)
{
    auto kern = stack_exact_kernel<MODE, W, LANES>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds_bytes, stream, args);
    return hipGetLastError();
}

template <int MODE, bool W>
static hipError_t launch_exact_lanes(StackArgs &args, int lanes, int grid, size_t lds_bytes,
                                     hipStream_t stream)
{
    switch (lanes) {
    case 64: return launch_exact<MODE, W, 64>(args, grid, lds_bytes, stream);
    case 32: return launch_exact<MODE, W, 32>(args, grid, lds_bytes, stream);
    case 16: return launch_exact<MODE, W, 16>(args, grid, lds_bytes, stream);
    default: return launch_exact<MODE, W, 4>(args, grid, lds_bytes, stream);
    }
}

int exact_plan(int mode, bool weighted, int n_frames, int n_pad, int max_lanes, int *lanes,
               size_t *lds_bytes)
{
    const int columns = ((mode == NL_ST_MAD_SIGMA) ||
                         (weighted && (mode == NL_ST_SIGMA || mode == NL_ST_WINSOR_SIGMA))) ? 2 : 1;
    const size_t n_alloc = (mode == NL_ST_LINEAR_FIT) ? (size_t)n_pad : (size_t)n_frames;
    for (int l = 64; l >= 4; l >>= 1) {
        if (l > max_lanes || l == 8) continue;
        const size_t bytes = (size_t)columns * n_alloc * l * sizeof(float);
        if (bytes <= kLdsBudgetBytes) {
            *lanes = l;
            *lds_bytes = bytes;
            return 0;
        }
    }
    return -1;
}

hipError_t launch_stack_exact(int mode, bool weighted, StackArgs &args, int lanes, int grid,
                              size_t lds_bytes, hipStream_t stream, const char **name)
{
    switch (mode) {
    case NL_ST_MEDIAN:
        *name = "stack_exact_kernel<median>";
        return launch_exact_lanes<NL_ST_MEDIAN, false>(args, lanes, grid, lds_bytes, stream);
    case NL_ST_SIGMA:
        if (weighted) {
            *name = "stack_exact_kernel<sigma,weighted>";
            return launch_exact_lanes<NL_ST_SIGMA, true>(args, lanes, grid, lds_bytes, stream);
        }
        *name = "stack_exact_kernel<sigma>";
        return launch_exact_lanes<NL_ST_SIGMA, false>(args, lanes, grid, lds_bytes, stream);
    case NL_ST_WINSOR_SIGMA:
        if (weighted) {
            *name = "stack_exact_kernel<winsor,weighted>";
            return launch_exact_lanes<NL_ST_WINSOR_SIGMA, true>(args, lanes, grid, lds_bytes, stream);
        }
        *name = "stack_exact_kernel<winsor>";
        return launch_exact_lanes<NL_ST_WINSOR_SIGMA, false>(args, lanes, grid, lds_bytes, stream);
    case NL_ST_MAD_SIGMA:
        *name = "stack_exact_kernel<mad>";
        return launch_exact_lanes<NL_ST_MAD_SIGMA, false>(args, lanes, grid, lds_bytes, stream);
    case NL_ST_LINEAR_FIT:
        *name = "stack_exact_kernel<linearfit>";
        return launch_exact_lanes<NL_ST_LINEAR_FIT, false>(args, lanes, grid, lds_bytes, stream);
    default:
        return hipErrorInvalidValue;
    }
}

hipError_t launch_reduce_counters(unsigned long long *partial, int n_blocks,
                                  unsigned long long *counters, hipStream_t stream, const unsigned *list_counts,
                                  int n_lists, int list_stride, bool zero_after)
{
    hipLaunchKernelGGL(reduce_counters_kernel, dim3(1), dim3(256), 0, stream, partial, n_blocks,
                       counters, list_counts, n_lists, list_stride, zero_after ? 1 : 0);
    return hipGetLastError();
}

}  // namespace nl
