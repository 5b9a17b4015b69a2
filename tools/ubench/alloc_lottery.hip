// Developer microbenchmark: does the streaming rate of a 128-frame stack depend on WHICH allocation it lives in?
// (median / mean 128 x 4096^2 run 8 % slower on about one handle in six.)  Eight (or argv[1], up to 28) allocations of
// 8 GiB (+ slack), the same streaming kernel over each (one lane per pixel, 128 nontemporal loads in flight, a sum), three
// rounds; prints the device address and the rate, and the rate again at a base shifted by 2 MiB and by 64 KiB.
//   hipcc --offload-arch=gfx950 -O3 alloc_lottery.hip -o alloc_lottery && ./alloc_lottery
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

__global__ __launch_bounds__(256) void stream128(const float *frames, float *out, long npix, long stride)
{
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    float v[128];
#pragma unroll
    for (int k = 0; k < 128; k++) v[k] = __builtin_nontemporal_load(frames + pix + (long)k * stride);
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int k = 0; k < 128; k += 4) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3]; }
    out[pix] = (s0 + s1) + (s2 + s3);
}

static float rate(const float *frames, float *out, long npix, long stride)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned grid = (unsigned)((npix + 255) / 256);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(stream128, dim3(grid), dim3(256), 0, 0, frames, out, npix, stride);
    hipEventRecord(e0);
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(stream128, dim3(grid), dim3(256), 0, 0, frames, out, npix, stride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return (float)((double)npix * 129 * 4 / (ms / 8) / 1e6);
}

int main(int argc, char **argv)
{
    const long npix = 4096L * 4096L, stride = npix + 16448;
    const size_t bytes = (size_t)stride * 128 * sizeof(float) + (8u << 20);
    float *out;
    hipMalloc(&out, npix * sizeof(float));
    float *buf[28];
    const int nalloc = argc > 1 ? atoi(argv[1]) : 8;
    for (int i = 0; i < nalloc; i++) {
        if (hipMalloc(&buf[i], bytes) != hipSuccess) { printf("no memory at %d\n", i); return 1; }
        hipMemset(buf[i], 0, bytes);
    }
    hipDeviceSynchronize();
    for (int round = 0; round < (nalloc > 8 ? 1 : 3); round++)
        for (int i = 0; i < nalloc; i++)
            printf("round %d allocation %d at %p: %7.1f GB/s   base + 2 MiB: %7.1f   base + 64 KiB: %7.1f\n", round, i, (void *)buf[i],
                   rate(buf[i], out, npix, stride), rate(buf[i] + (2u << 20) / 4, out, npix, stride), rate(buf[i] + (64u << 10) / 4, out, npix, stride));
    return 0;
}
