// Developer microbenchmark: sustained issue rate of v_min_f32/v_max_f32 (the
// sorting network's instruction mix) per SIMD on gfx950, vs waves per SIMD.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int REGS>
__global__ __launch_bounds__(256) void minmax_kernel(float *out, int iters)
{
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; i++) v[i] = (float)(threadIdx.x * 31 + i * 17 % 101);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i + 1 < REGS; i += 2) {
            const float lo = fminf(v[i], v[i + 1]), hi = fmaxf(v[i], v[i + 1]);
            v[i] = lo; v[i + 1] = hi;
        }
#pragma unroll
        for (int i = 1; i + 1 < REGS; i += 2) {
            const float lo = fminf(v[i], v[i + 1]), hi = fmaxf(v[i], v[i + 1]);
            v[i] = lo; v[i + 1] = hi;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < REGS; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int REGS>
__global__ __launch_bounds__(256) void fma_kernel(float *out, int iters)
{
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; i++) v[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < REGS; i++) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < REGS; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 4096 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    constexpr int R = 32;
    for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
        const int grid = 256 * blocks_per_cu;     // 256-thread blocks = 1 wave per SIMD each
        for (int which = 0; which < 2; which++) {
            if (which == 0) minmax_kernel<R><<<grid, 256>>>(d, 10); else fma_kernel<R><<<grid, 256>>>(d, 10);
            hipDeviceSynchronize();
            hipEventRecord(a);
            if (which == 0) minmax_kernel<R><<<grid, 256>>>(d, iters); else fma_kernel<R><<<grid, 256>>>(d, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double per_iter = which == 0 ? 2.0 * (R - 1) : (double)R;   // VALU instrs per lane per iter
            const double wave_instr = (double)grid * 4 * iters * per_iter;
            printf("%s waves/SIMD=%d: %.3f ms, %.3f T wave-instr/s chip, %.2f instr/ns/CU\n",
                   which == 0 ? "minmax" : "fma   ", blocks_per_cu, ms, wave_instr / ms / 1e9,
                   wave_instr / ms / 1e6 / 256);
        }
    }
    return 0;
}
