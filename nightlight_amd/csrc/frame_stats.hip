// frame_stats.hip -- whole-frame statistics and the 3x3 spatial median for gfx950.
// These replace the reference's only native code (three AVX2 assembly files):
//   calcMinMeanMaxAVX2 / calcVarianceAVX2   internal/stats/stats_amd64.s:28-143
//   estimateNoiseLineAVX2                   internal/stats/noise_amd64.s:78-195
//   medianFilterLine3x3AVX2                 internal/median/median3x3_amd64.s:62-236
// All are single-pass HBM streams (4 B read per sample): 16-byte loads,
// grid-stride, wave shuffle + LDS block reduction, one fp64 partial per
// workgroup, summed on the host (a few thousand values).
#include "stack_kernels.h"

namespace nl {

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// min / max in fp32, sum in fp64 (stats.go:264-277, stats_amd64.s:28-92).
// A comparison with NaN is false in the reference, so NaN never becomes
// min/max but poisons the sum: same here (explicit compares, not fmin/fmax).
__global__ __launch_bounds__(256) void min_sum_max_kernel(const float *data, int64_t n,
                                                           double *partial)
{
    float mn = data[0], mx = data[0];
    double sum = 0.0;
    const int64_t quads = n >> 2;
    const float4 *d4 = reinterpret_cast<const float4 *>(data);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = d4[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (e[j] < mn) mn = e[j];
            if (e[j] > mx) mx = e[j];
            sum += (double)e[j];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t i = quads << 2; i < n; i++) {
            const float e = data[i];
            if (e < mn) mn = e;
            if (e > mx) mx = e;
            sum += (double)e;
        }
    }
    __shared__ float s_mn[4], s_mx[4];
    __shared__ double s_sum[4];
    mn = wave_min_f(mn);
    mx = wave_max_f(mx);
    sum = wave_sum_d(sum);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mn[wave] = mn; s_mx[wave] = mx; s_sum[wave] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            mn = fminf(mn, s_mn[w]);
            mx = fmaxf(mx, s_mx[w]);
            sum += s_sum[w];
        }
        partial[3 * (size_t)blockIdx.x + 0] = (double)mn;
        partial[3 * (size_t)blockIdx.x + 1] = sum;
        partial[3 * (size_t)blockIdx.x + 2] = (double)mx;
    }
}

// sum of (double)(x - mean_fp32)^2 (stats.go:280-287, stats_amd64.s:102-143)
__global__ __launch_bounds__(256) void variance_kernel(const float *data, int64_t n, float mean,
                                                        double *partial)
{
    double sum = 0.0;
    const int64_t quads = n >> 2;
    const float4 *d4 = reinterpret_cast<const float4 *>(data);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = d4[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double d = (double)(e[j] - mean);
            sum += d * d;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t i = quads << 2; i < n; i++) {
            const double d = (double)(data[i] - mean);
            sum += d * d;
        }
    }
    __shared__ double s_sum[4];
    sum = wave_sum_d(sum);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// Immerkaer noise estimate (noise.go:32-55): per interior pixel the fp32
// convolution with [1,-2,1;-2,4,-2;1,-2,1] in the reference's tap order and
// without FMA (bit-identical per pixel), |.| accumulated in fp64 (the two
// reference paths -- pure Go and AVX2 -- already differ in summation order).
__global__ __launch_bounds__(256) void noise_kernel(const float *data, int width, int height,
                                                     double *partial)
{
    const int64_t iw = width - 2, ih = height - 2;
    const int64_t total = iw * ih;
    double sum = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = t / iw + 1, x = t % iw + 1;
        const float *r0 = data + (y - 1) * width + x;
        const float *r1 = r0 + width;
        const float *r2 = r1 + width;
        float conv = 0.0f;
        float pr;
        pr = r0[-1] * 1.0f;  conv += pr;
        pr = r0[0] * -2.0f;  conv += pr;
        pr = r0[1] * 1.0f;   conv += pr;
        pr = r1[-1] * -2.0f; conv += pr;
        pr = r1[0] * 4.0f;   conv += pr;
        pr = r1[1] * -2.0f;  conv += pr;
        pr = r2[-1] * 1.0f;  conv += pr;
        pr = r2[0] * -2.0f;  conv += pr;
        pr = r2[1] * 1.0f;   conv += pr;
        sum += (double)fabsf(conv);
    }
    __shared__ double s_sum[4];
    sum = wave_sum_d(sum);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// 19-step median-of-9 exchange network (median3x3.go:85-110); min/max only,
// so results are bit-exact for NaN-free input.
#define NL_CE(i, j) { const float lo_ = fminf(a##i, a##j); a##j = fmaxf(a##i, a##j); a##i = lo_; }
#define NL_MAXTO(i, j) { a##j = fmaxf(a##i, a##j); }
#define NL_MINTO(i, j) { a##i = fminf(a##i, a##j); }

__global__ __launch_bounds__(256) void median3x3_kernel(const float *in, float *out, int width,
                                                         int height)
{
    const int64_t total = (int64_t)width * height;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = t / width, x = t % width;
        if (y == 0 || y == height - 1 || x == 0 || x == width - 1) {
            out[t] = in[t];           // border rows / columns copied (median3x3.go:28-36)
            continue;
        }
        const float *r0 = in + t - width, *r1 = in + t, *r2 = in + t + width;
        float a0 = r0[-1], a1 = r0[0], a2 = r0[1];
        float a3 = r1[-1], a4 = r1[0], a5 = r1[1];
        float a6 = r2[-1], a7 = r2[0], a8 = r2[1];
        NL_CE(0, 1) NL_CE(3, 4) NL_CE(6, 7)
        NL_CE(1, 2) NL_CE(4, 5) NL_CE(7, 8)
        NL_CE(0, 1) NL_CE(3, 4) NL_CE(6, 7)
        NL_MAXTO(0, 3)
        NL_MAXTO(3, 6)
        NL_CE(1, 4)
        NL_MINTO(4, 7)
        NL_MAXTO(1, 4)
        NL_MINTO(5, 8)
        NL_MINTO(2, 5)
        NL_CE(2, 4)
        NL_MINTO(4, 6)
        NL_MAXTO(2, 4)
        out[t] = a4;
    }
}

// GatherAndMedian over every pixel = MedianFilter (internal/median/gather.go:26-38,
// internal/ops/pre/badpixels.go:54-77; masks from star/findstars.go:187-200, <= 21 offsets for
// the radii the reference uses, kMaskMax here).  One pixel per lane, the neighbourhood in
// registers.  MedianFloat32 (median3x3.go:115-119) is order independent, so instead of replaying
// quickselect the kernel ranks by counting: x_j is the k-th smallest iff #{x < x_j} <= k <
// #{x <= x_j}; odd count -> the middle value, even count -> 0.5 * (lower + upper) exactly as
// qsort.go:68-82 (the 9-value network of median3x3.go:85-110 returns the same middle value).
// Where part of the neighbourhood falls outside the data the reference's result depends on what
// earlier calls left in its scratch buffer (gather.go:37 takes the median of the WHOLE buffer);
// here it is the median of the values that exist -- the only history-free reading.
constexpr int kMaskMax = 32;
struct MaskArg { int off[kMaskMax]; };

__global__ __launch_bounds__(256) void median_mask_kernel(const float *in, float *out, int64_t n, MaskArg m, int len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[kMaskMax];
    int num = 0;
    bool has_nan = false;
#pragma unroll
    for (int j = 0; j < kMaskMax; j++) {
        const int64_t io = i + m.off[j];
        const bool ok = j < len && io >= 0 && io < n;
        x[j] = ok ? in[io] : __builtin_inff();          // missing: +Inf, never the k-th smallest for k < num
        has_nan |= x[j] != x[j];
        num += ok ? 1 : 0;
    }
    const int ku = num >> 1, kl = ku - 1;
    float upper = 0.0f, lower = 0.0f;
#pragma unroll
    for (int j = 0; j < kMaskMax; j++) {
        int lt = 0, le = 0;
#pragma unroll
        for (int t = 0; t < kMaskMax; t++) {
            lt += (x[t] < x[j]) ? 1 : 0;
            le += (x[t] <= x[j]) ? 1 : 0;
        }
        if (lt <= ku && ku < le) upper = x[j];
        if (lt <= kl && kl < le) lower = x[j];
    }
    float res = (num & 1) ? upper : 0.5f * (lower + upper);
    if (num == 0) res = __builtin_nanf("");                 // median3x3.go:116
    // The reference requires NaN-free input (median3x3.go:114: its quickselect does not terminate
    // properly on NaN).  Counting would match no rank and return a wrong value silently: a
    // neighbourhood holding a NaN yields NaN instead.
    if (has_nan) res = __builtin_nanf("");
    out[i] = res;
}

hipError_t launch_median_mask(const float *in, float *out, int64_t n, const int *mask, int len, hipStream_t stream)
{
    MaskArg m;
    for (int j = 0; j < kMaskMax; j++) m.off[j] = j < len ? mask[j] : 0;
    hipLaunchKernelGGL(median_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, n, m, len);
    return hipGetLastError();
}

hipError_t launch_min_sum_max(const float *data, int64_t n, double *partial, int blocks,
                              hipStream_t stream)
{
    hipLaunchKernelGGL(min_sum_max_kernel, dim3(blocks), dim3(256), 0, stream, data, n, partial);
    return hipGetLastError();
}

hipError_t launch_variance(const float *data, int64_t n, float mean, double *partial, int blocks,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(variance_kernel, dim3(blocks), dim3(256), 0, stream, data, n, mean, partial);
    return hipGetLastError();
}

hipError_t launch_noise(const float *data, int width, int height, double *partial, int blocks,
                        hipStream_t stream)
{
    hipLaunchKernelGGL(noise_kernel, dim3(blocks), dim3(256), 0, stream, data, width, height,
                       partial);
    return hipGetLastError();
}

hipError_t launch_median3x3(const float *in, float *out, int width, int height, hipStream_t stream)
{
    const int64_t total = (int64_t)width * height;
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(median3x3_kernel, dim3((int)g), dim3(256), 0, stream, in, out, width, height);
    return hipGetLastError();
}

}  // namespace nl
