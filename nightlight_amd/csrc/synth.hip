// synth.hip -- deterministic synthetic sub-exposure stack generated in HBM
// (SURVEY.md section 8d), so that 8-32 GiB benchmark inputs never cross PCIe.
// Counter-based RNG keyed by (seed, frame, pixel-of-the-full-image): a row tile
// holds exactly the rows of the whole image's stack, whatever the sharding.
#include "stack_kernels.h"

namespace nl {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float u01(uint32_t bits)   // (0,1]
{
    return ((float)(bits >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void fill_synthetic_kernel(float *frames, int64_t stride,
                                                              int n_frames, int width, int height,
                                                              int row0, int rows, uint64_t seed)
{
    const int64_t tile_px = (int64_t)rows * width;
    const int64_t total = tile_px * n_frames;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx / tile_px);
        const int64_t t = idx - (int64_t)k * tile_px;
        const int y = row0 + (int)(t / width);
        const int x = (int)(t % width);
        const int64_t gp = (int64_t)y * width + x;          // pixel of the full image

        const uint64_t h0 = splitmix64(seed ^ splitmix64(((uint64_t)k << 40) ^ (uint64_t)gp));
        const uint64_t h1 = splitmix64(h0);
        const uint64_t h2 = splitmix64(h1);

        // per-frame sky level, gain and noise (distinct noise => distinct weights)
        const float bg = 1000.0f + 5.0f * __sinf((float)k);
        const float gain = 1.0f + 0.02f * __cosf(1.7f * (float)k);
        const float sigma = 30.0f * (1.0f + 0.5f * (float)(k % 7) / 6.0f);
        // smooth sky gradient in [0,200]
        const float sky = 200.0f * (0.5f * (float)x / (float)width + 0.5f * (float)y / (float)height);
        // Box-Muller
        const float u1 = u01((uint32_t)h0), u2 = u01((uint32_t)(h0 >> 32));
        const float g = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
        float v = bg + gain * sky + sigma * g;
        // outliers: 0.4 % hot (cosmic ray / satellite), 0.1 % cold
        const float uo = u01((uint32_t)h1), um = u01((uint32_t)(h1 >> 32));
        if (uo < 0.004f) v += 300.0f + 19700.0f * um;
        else if (uo < 0.005f) v -= 100.0f + 800.0f * um;
        // alignment out-of-bounds: NaN rows at the top, NaN columns at the right
        if (y < (k % 9) || x >= width - (k % 5)) v = __builtin_nanf("");
        // one 8x8 patch with no data in any frame (exercises RefFrameLoc)
        const int py = height / 2, px = width / 2;
        if (y >= py && y < py + 8 && x >= px && x < px + 8) v = __builtin_nanf("");
        (void)h2;
        frames[(int64_t)k * stride + t] = v;
    }
}

hipError_t launch_fill_synthetic(float *frames, int64_t stride, int n_frames, int width,
                                 int height, int row0, int rows, uint64_t seed,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3(256 * 16), dim3(256), 0, stream, frames, stride,
                       n_frames, width, height, row0, rows, seed);
    return hipGetLastError();
}

}  // namespace nl
