// ingest.hip -- the data formats and per-frame steps either side of the stack
// (SURVEY.md section 8f rows F3 / F4), so that a frame can go from its on-disk
// bytes to its slot of the planar [N][rows*W] stack buffer without a CPU pass:
//   fits_decode_kernel   internal/fits/read.go:172-445   big-endian BITPIX 8/16/32/64/-32/-64
//                        -> fp32, v = float32(val)*BSCALE + BZERO, + min / max / sum
//   fits_encode_kernel   internal/fits/write.go:182-200  fp32 -> big-endian, NaN -> 0
//   affine_kernel        internal/fits/pixelops.go:601-605  MatchHistogram x*m + o
//   project_kernel       internal/fits/project.go:26-76  bilinear resampling through the
//                        inverted Transform2D (internal/star/coord.go:141-145, 159-199),
//                        out of bounds -> the given value (NaN in the pipeline)
// All of it is elementwise fp32 with the reference's operation order (no FMA:
// -ffp-contract=off), hence bit-exact; all of it is HBM-bound.
#include <float.h>

#include "stack_kernels.h"

namespace nl {

namespace {

__device__ __forceinline__ unsigned bswap32(unsigned x) { return __builtin_bswap32(x); }

// one decoded value: bytes of element i of a big-endian payload -> float32(val)
template <int BITPIX>
__device__ __forceinline__ float decode_one(const unsigned char *raw, int64_t i)
{
    if constexpr (BITPIX == 8) {
        return (float)raw[i];                                                       // read.go:192-193
    } else if constexpr (BITPIX == 16) {
        const unsigned short u = reinterpret_cast<const unsigned short *>(raw)[i];
        return (float)(short)(unsigned short)((u << 8) | (u >> 8));                 // read.go:234
    } else if constexpr (BITPIX == 32) {
        return (float)(int)bswap32(reinterpret_cast<const unsigned *>(raw)[i]);
    } else if constexpr (BITPIX == 64) {
        const unsigned long long u = __builtin_bswap64(reinterpret_cast<const unsigned long long *>(raw)[i]);
        return (float)(long long)u;                                                 // read.go:325-327
    } else if constexpr (BITPIX == -32) {
        return __uint_as_float(bswap32(reinterpret_cast<const unsigned *>(raw)[i])); // read.go:372-373
    } else {
        const unsigned long long u = __builtin_bswap64(reinterpret_cast<const unsigned long long *>(raw)[i]);
        return (float)__longlong_as_double((long long)u);                           // read.go:421-423
    }
}

// four consecutive values with one wide load (the payload base is 16-byte aligned)
template <int BITPIX>
__device__ __forceinline__ void decode_four(const unsigned char *raw, int64_t i4, float (&v)[4])
{
    if constexpr (BITPIX == 8) {
        const unsigned w = reinterpret_cast<const unsigned *>(raw)[i4];
        v[0] = (float)(w & 0xffu); v[1] = (float)((w >> 8) & 0xffu);
        v[2] = (float)((w >> 16) & 0xffu); v[3] = (float)(w >> 24);
    } else if constexpr (BITPIX == 16) {
        const uint2 w = reinterpret_cast<const uint2 *>(raw)[i4];
        const unsigned a = bswap32(w.x), b = bswap32(w.y);        // bytes b0 b1 b2 b3 -> (b0b1)(b2b3)
        v[0] = (float)(short)(a >> 16); v[1] = (float)(short)(a & 0xffffu);
        v[2] = (float)(short)(b >> 16); v[3] = (float)(short)(b & 0xffffu);
    } else if constexpr (BITPIX == 32 || BITPIX == -32) {
        const uint4 w = reinterpret_cast<const uint4 *>(raw)[i4];
        const unsigned u[4] = {bswap32(w.x), bswap32(w.y), bswap32(w.z), bswap32(w.w)};
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = BITPIX == 32 ? (float)(int)u[j] : __uint_as_float(u[j]);
    } else {
        const uint4 w0 = reinterpret_cast<const uint4 *>(raw)[2 * i4];
        const uint4 w1 = reinterpret_cast<const uint4 *>(raw)[2 * i4 + 1];
        const unsigned long long u[4] = {
            ((unsigned long long)bswap32(w0.x) << 32) | bswap32(w0.y),
            ((unsigned long long)bswap32(w0.z) << 32) | bswap32(w0.w),
            ((unsigned long long)bswap32(w1.x) << 32) | bswap32(w1.y),
            ((unsigned long long)bswap32(w1.z) << 32) | bswap32(w1.w)};
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = BITPIX == 64 ? (float)(long long)u[j] : (float)__longlong_as_double((long long)u[j]);
    }
}

// partial[3*b + {0,1,2}] = {min, max, sum} of block b, as doubles (min / max are exact floats)
__device__ __forceinline__ void block_min_max_sum(float mn, float mx, double sum, double *partial)
{
    __shared__ float s_mn[4], s_mx[4];
    __shared__ double s_sum[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        sum += __shfl_xor(sum, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; s_sum[threadIdx.x >> 6] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[3 * blockIdx.x + 0] = (double)fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        partial[3 * blockIdx.x + 1] = (double)fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        partial[3 * blockIdx.x + 2] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
}

// `if v < min` / `if v > max` of the reference never fire for NaN: same as fminf / fmaxf
template <int BITPIX, bool AFFINE>
__global__ __launch_bounds__(256) void fits_decode_kernel(const unsigned char *raw, int64_t n, float bscale,
                                                          float bzero, float mult, float off, float *out,
                                                          double *partial)
{
    float mn = FLT_MAX, mx = -FLT_MAX;
    double sum = 0.0;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
        float v[4];
        decode_four<BITPIX>(raw, i4, v);
        float4 o;
        float *po = &o.x;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float x = v[j] * bscale + bzero;                       // read.go:193 (mul, then add)
            mn = fminf(mn, x); mx = fmaxf(mx, x);
            sum += (double)x;
            po[j] = AFFINE ? x * mult + off : x;                         // pixelops.go:604
        }
        reinterpret_cast<float4 *>(out)[i4] = o;
    }
    // tail (n not a multiple of 4): first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float x = decode_one<BITPIX>(raw, i) * bscale + bzero;
        mn = fminf(mn, x); mx = fmaxf(mx, x);
        sum += (double)x;
        out[i] = AFFINE ? x * mult + off : x;
    }
    block_min_max_sum(mn, mx, sum, partial);
}

__global__ __launch_bounds__(256) void fits_encode_kernel(const float *data, int64_t n, int replace_nans,
                                                          unsigned *raw)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = data[i];
        if (replace_nans && d != d) d = 0.0f;                            // write.go:191
        raw[i] = bswap32(__float_as_uint(d));
    }
}

__global__ __launch_bounds__(256) void affine_kernel(float *data, int64_t n, float mult, float off)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
        float4 x = reinterpret_cast<float4 *>(data)[i4];
        x.x = x.x * mult + off; x.y = x.y * mult + off; x.z = x.z * mult + off; x.w = x.w * mult + off;
        reinterpret_cast<float4 *>(data)[i4] = x;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        data[i] = data[i] * mult + off;
    }
}

// One thread per destination pixel of the tile rows [row0, row0+rows).  A wave
// covers 64 consecutive columns of one row, so its four source taps are four
// (nearly) contiguous row segments: coalesced, and the second row of taps is
// the next wave-row's first -- the L2 serves it.
template <bool AFFINE>
__global__ __launch_bounds__(256) void project_kernel(const float *src, int src_w, int src_h, float *dst,
                                                      int dst_w, int row0, int rows, float ia, float ib,
                                                      float ic, float id, float ie, float iff, float oob,
                                                      float mult, float off)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (col >= dst_w || r >= rows) return;
    const float px = (float)col, py = (float)(row0 + r);
    const float X = ia * px + ib * py + ic;                              // coord.go:142 (left to right)
    const float Y = id * px + ie * py + iff;                             // coord.go:143
    const float fx = floorf(X), fy = floorf(Y);                          // project.go:52
    // int32(math.Floor(x)) of NaN / out-of-range is negative on amd64 => out of bounds (project.go:56)
    bool ok = fx >= 0.0f && fy >= 0.0f && fx < 2147483520.0f && fy < 2147483520.0f;
    int xl = 0, yl = 0;
    if (ok) {
        xl = (int)fx; yl = (int)fy;
        ok = (int64_t)xl + 1 < src_w && (int64_t)yl + 1 < src_h;
    }
    float v = oob;
    if (ok) {
        const float xr = X - (float)xl, yr = Y - (float)yl;              // project.go:54
        const int64_t p = (int64_t)xl + (int64_t)yl * src_w;
        const float omx = 1.0f - xr, omy = 1.0f - yr;
        const float vyl = src[p] * omx + src[p + 1] * xr;                // project.go:68
        const float vyh = src[p + src_w] * omx + src[p + src_w + 1] * xr;
        v = vyl * omy + vyh * yr;                                        // project.go:70
        if (AFFINE) v = v * mult + off;
    } else if (AFFINE) {
        v = oob * mult + off;                                            // MatchHistogram runs over every pixel
    }
    dst[(int64_t)r * dst_w + col] = v;
}

}  // namespace

template <int BITPIX>
static void launch_decode_t(const void *raw, int64_t n, float bscale, float bzero, bool affine, float mult,
                            float off, float *out, double *partial, int blocks, hipStream_t stream)
{
    const unsigned char *r = static_cast<const unsigned char *>(raw);
    if (affine)
        hipLaunchKernelGGL((fits_decode_kernel<BITPIX, true>), dim3(blocks), dim3(256), 0, stream, r, n, bscale,
                           bzero, mult, off, out, partial);
    else
        hipLaunchKernelGGL((fits_decode_kernel<BITPIX, false>), dim3(blocks), dim3(256), 0, stream, r, n, bscale,
                           bzero, mult, off, out, partial);
}

int fits_bytes_per_value(int bitpix)
{
    switch (bitpix) {
    case 8: return 1;
    case 16: return 2;
    case 32: case -32: return 4;
    case 64: case -64: return 8;
    default: return 0;
    }
}

hipError_t launch_fits_decode(const void *raw, int bitpix, int64_t n, float bscale, float bzero, bool affine,
                              float mult, float off, float *out, double *partial, int blocks,
                              hipStream_t stream)
{
    switch (bitpix) {
    case 8:   launch_decode_t<8>(raw, n, bscale, bzero, affine, mult, off, out, partial, blocks, stream); break;
    case 16:  launch_decode_t<16>(raw, n, bscale, bzero, affine, mult, off, out, partial, blocks, stream); break;
    case 32:  launch_decode_t<32>(raw, n, bscale, bzero, affine, mult, off, out, partial, blocks, stream); break;
    case 64:  launch_decode_t<64>(raw, n, bscale, bzero, affine, mult, off, out, partial, blocks, stream); break;
    case -32: launch_decode_t<-32>(raw, n, bscale, bzero, affine, mult, off, out, partial, blocks, stream); break;
    case -64: launch_decode_t<-64>(raw, n, bscale, bzero, affine, mult, off, out, partial, blocks, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fits_encode(const float *data, int64_t n, int replace_nans, void *raw, hipStream_t stream)
{
    const int64_t want = (n + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
    hipLaunchKernelGGL(fits_encode_kernel, dim3(blocks), dim3(256), 0, stream, data, n, replace_nans,
                       static_cast<unsigned *>(raw));
    return hipGetLastError();
}

hipError_t launch_affine(float *data, int64_t n, float mult, float off, hipStream_t stream)
{
    const int64_t want = ((n >> 2) + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
    hipLaunchKernelGGL(affine_kernel, dim3(blocks), dim3(256), 0, stream, data, n, mult, off);
    return hipGetLastError();
}

hipError_t launch_project(const float *src, int src_w, int src_h, float *dst, int dst_w, int row0, int rows,
                          const float inv[6], float oob, bool affine, float mult, float off,
                          hipStream_t stream)
{
    const dim3 grid((unsigned)((dst_w + 255) / 256), (unsigned)rows);
    if (affine)
        hipLaunchKernelGGL((project_kernel<true>), grid, dim3(256), 0, stream, src, src_w, src_h, dst, dst_w, row0,
                           rows, inv[0], inv[1], inv[2], inv[3], inv[4], inv[5], oob, mult, off);
    else
        hipLaunchKernelGGL((project_kernel<false>), grid, dim3(256), 0, stream, src, src_w, src_h, dst, dst_w, row0,
                           rows, inv[0], inv[1], inv[2], inv[3], inv[4], inv[5], oob, mult, off);
    return hipGetLastError();
}

}  // namespace nl
