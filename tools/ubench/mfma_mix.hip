// Developer microbenchmark: does v_mfma_f32_4x4x1_16b_f32 beside a VALU-bound instruction stream cost VALU issue slots?
// Per iteration every wave issues 64 v_min_f32 (half rate, the sorting networks' instruction) plus, per variant,
//   none | 16 v_fmac_f32 + 16 v_add_f32 | 32 v_mfma_f32_4x4x1 (two accumulator quads, alternating) | 16 fmac + 16 mfma
// grid = 256 * waves-per-SIMD workgroups of 256 threads (one wave per SIMD each).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters)
{
    float r[16];
    for (int i = 0; i < 16; i++) r[i] = (float)(threadIdx.x * 7 + i);
    float a = (float)threadIdx.x + 1.0f, b = 0.5f;
    float e[16];
    for (int i = 0; i < 16; i++) e[i] = (float)(threadIdx.x & 15) * 0.25f + (float)i;
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    float q0 = 0, q1 = 0, d0 = 0, d1 = 0;
    const float one = 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                if (VARIANT == 1 && (i & 3) == 0) {
                    const int k = 4 * rep + i / 4;
                    asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(k & 1 ? q1 : q0) : "v"(e[k]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(k & 1 ? d1 : d0) : "v"(e[k]));
                }
                if (VARIANT == 2 && (i & 3) == 0) {
                    const int k = 4 * rep + i / 4;
                    if (k & 1) {
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(e[k], e[k], acc1, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(one, e[k], acc3, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(e[k], e[k], acc0, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(one, e[k], acc2, 0, 0, 0);
                    }
                }
                if (VARIANT == 3 && (i & 3) == 0) {
                    const int k = 4 * rep + i / 4;
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(k & 1 ? d1 : d0) : "v"(e[k]));
                    if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(e[k], e[k], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(e[k], e[k], acc0, 0, 0, 0);
                }
            }
        }
        asm volatile("" : "+v"(b));
    }
    float s = q0 + q1 + d0 + d1;
    for (int i = 0; i < 16; i++) s += r[i];
    for (int i = 0; i < 4; i++) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// semantics check: lane l of a 4-lane block gets D[i][l] in register i; A = B = e  ->  register (l % 4) holds sum e_l^2
__global__ void k_sem(float *out)
{
    f4 acc = {0, 0, 0, 0}, accs = {0, 0, 0, 0};
    const float e1 = (float)(threadIdx.x + 1), e2 = 0.5f * (float)threadIdx.x;
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(e1, e1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(e2, e2, acc, 0, 0, 0);
    accs = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, e1, accs, 0, 0, 0);
    accs = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, e2, accs, 0, 0, 0);
    for (int i = 0; i < 4; i++) { out[threadIdx.x * 8 + i] = acc[i]; out[threadIdx.x * 8 + 4 + i] = accs[i]; }
}

template <int V>
static double run(float *d, int wps, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * wps;
    k_mix<V><<<grid, 256>>>(d, 5);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_mix<V><<<grid, 256>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 2048 * sizeof(float));
    float h[64 * 8];
    k_sem<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const float e1 = (float)(l + 1), e2 = 0.5f * (float)l;
        const float want_q = e1 * e1 + e2 * e2, want_d = e1 + e2;
        if (h[l * 8 + (l & 3)] != want_q) bad++;
        for (int i = 0; i < 4; i++) if (h[l * 8 + 4 + i] != want_d) bad++;
    }
    printf("semantics: %s (lane 5: q regs %g %g %g %g, d regs %g %g %g %g)\n", bad ? "MISMATCH" : "ok", h[40], h[41], h[42], h[43], h[44], h[45], h[46], h[47]);
    const int iters = 2000;
    for (int wps : {1, 2, 3, 4}) {
        const double t0 = run<0>(d, wps, iters), t1 = run<1>(d, wps, iters), t2 = run<2>(d, wps, iters), t3 = run<3>(d, wps, iters);
        printf("waves/SIMD=%d: 64 min: %.3f ms | + 16 fmac + 16 add: %.3f ms | + 32 mfma 4x4x1: %.3f ms | + 16 add + 16 mfma: %.3f ms\n", wps, t0, t1, t2, t3);
    }
    return 0;
}
