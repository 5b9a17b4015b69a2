// stack_mean.hip -- StackMean / StackMeanWeighted (internal/ops/stack/stack.go:307-366)
// and the stack-of-stacks axpy (StackIncremental, stack.go:924-944) for gfx950.
//
// Pure streaming: each lane owns 4 consecutive pixels (16-byte loads, 1 KiB per
// wave instruction), walks the frames in frame order -- the reference's
// summation order, so results are bit-exact -- with 8 frames of loads in
// flight.  No LDS, no reduction across lanes.  HBM-bound: 4*(N+1) bytes/pixel.
#include "stack_kernels.h"

namespace nl {

template <bool W>
__device__ __forceinline__ void mean_step(float v, float w, float &sum, float &wsum, int &cnt)
{
    if (v == v) {                 // math.IsNaN only: +-Inf are data
        if (W) {
            const float p = v * w;
            sum += p;
            wsum += w;
        } else {
            sum += v;
        }
        cnt++;
    }
}

template <bool W>
__global__ __launch_bounds__(256) void stack_mean_vec4_kernel(StackArgs p)
{
    const int64_t quads = p.npix >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        const float4 *fr = reinterpret_cast<const float4 *>(p.frames) + q;
        const int64_t stride4 = p.stride >> 2;
        float s[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0};
        int c[4] = {0, 0, 0, 0};
        int k = 0;
        for (; k + 8 <= p.n_frames; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&fr[(int64_t)(k + u) * stride4]));
                v[u] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float w = W ? p.weights[k + u] : 0.0f;
                mean_step<W>(v[u].x, w, s[0], ws[0], c[0]);
                mean_step<W>(v[u].y, w, s[1], ws[1], c[1]);
                mean_step<W>(v[u].z, w, s[2], ws[2], c[2]);
                mean_step<W>(v[u].w, w, s[3], ws[3], c[3]);
            }
        }
        for (; k < p.n_frames; k++) {
            const float4 v = fr[(int64_t)k * stride4];
            const float w = W ? p.weights[k] : 0.0f;
            mean_step<W>(v.x, w, s[0], ws[0], c[0]);
            mean_step<W>(v.y, w, s[1], ws[1], c[1]);
            mean_step<W>(v.z, w, s[2], ws[2], c[2]);
            mean_step<W>(v.w, w, s[3], ws[3], c[3]);
        }
        float4 r;
        float *rr = reinterpret_cast<float *>(&r);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float den = W ? ws[j] : (float)c[j];
            rr[j] = (c[j] == 0) ? p.ref_loc : s[j] / den;
        }
        {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 rv = {r.x, r.y, r.z, r.w};
            NL_STORE_RESULT(reinterpret_cast<f4 *>(p.out) + q, rv);
        }
    }
}

// scalar variant: tail pixels, or tiles whose stride is not a multiple of 4
template <bool W>
__global__ __launch_bounds__(256) void stack_mean_scalar_kernel(StackArgs p, int64_t first)
{
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.npix;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0, ws = 0;
        int c = 0;
        for (int k = 0; k < p.n_frames; k++) {
            const float v = p.frames[(int64_t)k * p.stride + i];
            mean_step<W>(v, W ? p.weights[k] : 0.0f, s, ws, c);
        }
        const float den = W ? ws : (float)c;
        p.out[i] = (c == 0) ? p.ref_loc : s / den;
    }
}

static int grid_for(int64_t items, int block, int cap)
{
    int64_t g = (items + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_stack_mean(bool weighted, const StackArgs &args, hipStream_t stream,
                             const char **name)
{
    const bool vec_ok = (args.stride % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(args.frames) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(args.out) & 15) == 0);
    int64_t done = 0;
    *name = weighted ? "stack_mean_vec4_kernel<weighted>" : "stack_mean_vec4_kernel";
    if (vec_ok && args.npix >= 4) {
        const int64_t quads = args.npix >> 2;
        const int grid = grid_for(quads, 256, 256 * 64);
        if (weighted)
            hipLaunchKernelGGL(stack_mean_vec4_kernel<true>, dim3(grid), dim3(256), 0, stream, args);
        else
            hipLaunchKernelGGL(stack_mean_vec4_kernel<false>, dim3(grid), dim3(256), 0, stream, args);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done = quads << 2;
    }
    if (done < args.npix) {
        const int grid = grid_for(args.npix - done, 256, 256 * 64);
        if (weighted)
            hipLaunchKernelGGL(stack_mean_scalar_kernel<true>, dim3(grid), dim3(256), 0, stream, args, done);
        else
            hipLaunchKernelGGL(stack_mean_scalar_kernel<false>, dim3(grid), dim3(256), 0, stream, args, done);
        return hipGetLastError();
    }
    return hipSuccess;
}

// StackIncremental (stack.go:924-937): acc = x*w (first) or acc += x*w
__global__ __launch_bounds__(256) void axpy_kernel(float *acc, const float *x, float w, int first,
                                                   int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float p = x[i] * w;
        acc[i] = first ? p : acc[i] + p;
    }
}

// StackIncrementalFinalize (stack.go:940-943)
__global__ __launch_bounds__(256) void scale_kernel(float *acc, float factor, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        acc[i] = acc[i] * factor;
}

hipError_t launch_axpy(float *acc, const float *x, float weight, int first, int64_t n,
                       hipStream_t stream)
{
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 256, 256 * 32)), dim3(256), 0, stream, acc, x,
                       weight, first, n);
    return hipGetLastError();
}

hipError_t launch_scale(float *acc, float factor, int64_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256, 256 * 32)), dim3(256), 0, stream, acc,
                       factor, n);
    return hipGetLastError();
}

}  // namespace nl
