// Developer microbenchmark: how fast can a wave stream N frames of a pixel column under different frame layouts?
// (the dense 2^26-byte frame stride cost the 512-frame kernel 12 %.)  Every lane owns a pixel (or LPP
// lanes share one, as the 129 ... 512-frame kernels do), loads its N samples with all loads in flight, adds them up and writes
// the sum -- the access pattern of the stacking kernels without their arithmetic.
//   layouts: planar, frame stride = pixels + pad floats;   tiled: blocks of B pixels x N frames contiguous.
//   hipcc --offload-arch=gfx950 -O3 frame_layout.hip -o frame_layout && ./frame_layout
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int N, int LPP, bool NT, bool ROWS = false>
__global__ __launch_bounds__(256) void stream(const float *frames, float *out, long npix, long frame_stride, long pixel_block, long block_stride)
{
    // sample k of pixel p: frames[(p / pixel_block) * block_stride + k * frame_stride + p % pixel_block]
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    // ROWS: the wave's 64 / LPP pixels sit in consecutive lanes, role = lane / (64 / LPP) -- every row of lanes reads one frame
    const int lane = threadIdx.x & 63;
    const long pix = ROWS ? (t / 64) * (64 / LPP) + lane % (64 / LPP) : t / LPP;
    const int role = ROWS ? lane / (64 / LPP) : (int)(t % LPP);
    if (pix >= npix) return;
    const float *base = frames + (pix / pixel_block) * block_stride + pix % pixel_block;
    float v[N / LPP];
#pragma unroll
    for (int k = 0; k < N / LPP; k++) { const float *a = base + (long)(k * LPP + role) * frame_stride; v[k] = NT ? __builtin_nontemporal_load(a) : *a; }
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int k = 0; k < N / LPP; k += 4) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3]; }
    float s = (s0 + s1) + (s2 + s3);
    if (LPP > 1) s += __shfl_xor(s, ROWS ? 64 / LPP : 1);
    if (LPP > 2) s += __shfl_xor(s, ROWS ? 32 : 2);
    if (role == 0) out[pix] = s;
}

template <int N, int LPP, bool NT, bool ROWS = false>
static void run(const char *name, float *frames, float *out, long npix, long frame_stride, long pixel_block, long block_stride)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned grid = (unsigned)((npix * LPP + 255) / 256);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((stream<N, LPP, NT, ROWS>), dim3(grid), dim3(256), 0, 0, frames, out, npix, frame_stride, pixel_block, block_stride);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((stream<N, LPP, NT, ROWS>), dim3(grid), dim3(256), 0, 0, frames, out, npix, frame_stride, pixel_block, block_stride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("N %3d lanes/pixel %d %s  %-34s %8.3f ms  %7.1f GB/s\n", N, LPP, ROWS ? (NT ? "nt rows" : "pl rows") : (NT ? "nt   " : "plain"), name, ms, (double)npix * (N + 1) * 4 / ms / 1e6);
    fflush(stdout);
}

template <int N, int LPP, bool NT, bool ROWS = false>
static void layouts(float *frames, float *out, long npix)
{
    char name[64];
    for (long pad : {0L, 1040L, 4160L, 16448L, 32832L}) {
        snprintf(name, sizeof name, "planar, stride + %ld floats", pad);
        run<N, LPP, NT, ROWS>(name, frames, out, npix, npix + pad, npix, 0);
    }
    for (long b : {64L, 256L, 1024L, 16384L}) {
        snprintf(name, sizeof name, "tiled, blocks of %ld pixels", b);
        run<N, LPP, NT, ROWS>(name, frames, out, npix, b, b, b * N);
    }
}

int main()
{
    const long npix = 4096L * 4096L;
    float *frames, *out;
    const size_t bytes = (size_t)(npix + 32832) * 512 * sizeof(float);
    if (hipMalloc(&frames, bytes) != hipSuccess || hipMalloc(&out, npix * sizeof(float)) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMemset(frames, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 1; rep++) {
        layouts<128, 1, true>(frames, out, npix);
        layouts<512, 4, true>(frames, out, npix);
        layouts<512, 4, true, true>(frames, out, npix);
        layouts<512, 4, false, true>(frames, out, npix);
        layouts<256, 2, true>(frames, out, npix);
        layouts<256, 2, true, true>(frames, out, npix);
    }
    return 0;
}
