// Developer microbenchmark: how many workgroups of a given size / LDS footprint are resident per CU on gfx950.
// Every workgroup sleeps a fixed wall time (s_memrealtime, 100 MHz); a grid of G workgroups then takes
// ceil(G / resident) sleeps.   hipcc --offload-arch=gfx950 -O3 resident.hip -o resident && ./resident
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void sleeper(int ticks, int *sink)
{
    extern __shared__ int lds[];
    if (threadIdx.x == 0) lds[0] = ticks;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < lds[0]) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) sink[0] = lds[1];
}

int main()
{
    int *sink;
    hipMalloc(&sink, 4);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const int ticks = 20000;                                  // 200 us
    for (int threads : {64, 128, 256}) {
        for (size_t lds : {(size_t)0, (size_t)1536, (size_t)4096, (size_t)6144, (size_t)8192}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int grid = cus * 64;
            hipLaunchKernelGGL(sleeper, dim3(grid), dim3(threads), lds, 0, 10, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(sleeper, dim3(grid), dim3(threads), lds, 0, ticks, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double gens = ms / (ticks / 1e5);
            printf("threads %3d lds %5zu: %.3f ms = %.2f sleeps -> about %.1f workgroups (%.1f waves) resident per CU\n", threads, lds, ms,
                   gens, 64.0 / gens, 64.0 / gens * threads / 64);
        }
    }
    return 0;
}
