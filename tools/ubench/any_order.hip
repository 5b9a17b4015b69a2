// Developer microbenchmark: does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on gfx950?
// Two small sleeping kernels back to back: in order they take two sleeps, any-order one.  (hip_ext.h: "not supported on AMD
// GFX9xx boards" -- accepted and ignored: 0.207 / 0.206 ms with the flag off / on, profiles/r05_any_order_ubench.txt.)
//   hipcc --offload-arch=gfx950 -O3 any_order.hip -o any_order && ./any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"

__global__ void sleeper(int ticks, int *sink)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) sink[0] = 1;
}

int main()
{
    int *sink;
    hipMalloc(&sink, 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int ticks = 10000;          // 100 us at 100 MHz
    for (unsigned flags : {0u, (unsigned)hipExtAnyOrderLaunch, 0u, (unsigned)hipExtAnyOrderLaunch}) {
        hipLaunchKernelGGL(sleeper, dim3(8), dim3(64), 0, s, 10, sink);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(sleeper, dim3(8), dim3(64), 0, s, ticks, sink);
        hipExtLaunchKernelGGL(sleeper, dim3(8), dim3(64), 0, s, nullptr, nullptr, flags, ticks, sink);
        hipLaunchKernelGGL(sleeper, dim3(8), dim3(64), 0, s, 10, sink);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("second kernel flags %u: %.3f ms for two 0.100 ms sleepers (%s)\n", flags, ms, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
