// Developer microbenchmark: host cost of the HIP runtime calls a handle's create / destroy is made of.
//   hipcc --offload-arch=gfx950 -O3 runtime_call_cost.hip -o runtime_call_cost && ./runtime_call_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    void *w; hipMalloc(&w, 1 << 20); hipFree(w);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); hipStream_t s[8]; for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        double t1 = now(); for (auto &x : s) hipStreamDestroy(x);
        double t2 = now(); void *p[8]; for (auto &x : p) hipMalloc(&x, 4096);
        double t3 = now(); for (auto &x : p) hipFree(x);
        double t4 = now(); hipEvent_t e[8]; for (auto &x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
        double t5 = now(); for (auto &x : e) hipEventDestroy(x);
        double t6 = now(); void *big; hipMalloc(&big, (size_t)1 << 30);
        double t7 = now(); hipFree(big);
        double t8 = now();
        printf("per call (us): stream create %.1f destroy %.1f | malloc 4 KiB %.1f free %.1f | event create %.1f destroy %.1f | malloc 1 GiB %.1f free %.1f\n",
               (t1 - t0) / 8, (t2 - t1) / 8, (t3 - t2) / 8, (t4 - t3) / 8, (t5 - t4) / 8, (t6 - t5) / 8, t7 - t6, t8 - t7);
    }
    return 0;
}
