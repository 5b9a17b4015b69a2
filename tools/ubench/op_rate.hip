// Developer microbenchmark: sustained issue rate of individual VALU opcodes on
// gfx950 (independent instructions, 16 destination registers per lane).
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP2(name, asmstr)                                                                    \
    __global__ __launch_bounds__(256) void k_##name(unsigned *out, int iters)                  \
    {                                                                                        \
        unsigned r[16];                                                                      \
        for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 7 + i * 1315423911u;               \
        unsigned a = threadIdx.x + 1, b = threadIdx.x * 3 + 5;                               \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int rep = 0; rep < 4; rep++) {                             \
                _Pragma("unroll") for (int i = 0; i < 16; i++)                                \
                    asm volatile(asmstr : "+v"(r[i]) : "v"(a), "v"(b));                      \
            }                                                                                \
        }                                                                                    \
        unsigned s = 0;                                                                      \
        for (int i = 0; i < 16; i++) s += r[i];                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }

OP2(min_f32, "v_min_f32 %0, %0, %1")
OP2(max_f32, "v_max_f32 %0, %0, %1")
OP2(min_i32, "v_min_i32 %0, %0, %1")
OP2(min_u32, "v_min_u32 %0, %0, %1")
OP2(add_f32, "v_add_f32 %0, %0, %1")
OP2(add_u32, "v_add_u32 %0, %0, %1")
OP2(fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP2(min3_f32, "v_min3_f32 %0, %0, %1, %2")
OP2(med3_f32, "v_med3_f32 %0, %0, %1, %2")
OP2(min3_i32, "v_min3_i32 %0, %0, %1, %2")
OP2(med3_i32, "v_med3_i32 %0, %0, %1, %2")
OP2(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP2(cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1")
OP2(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1")
OP2(pk_min_u16, "v_pk_min_u16 %0, %0, %1")
OP2(pk_min_f16, "v_pk_min_f16 %0, %0, %1")
OP2(xor_b32, "v_xor_b32 %0, %0, %1")
OP2(mov_b32, "v_mov_b32 %0, %1")
OP2(perm_b32, "v_perm_b32 %0, %0, %1, %2")
OP2(alignbit, "v_alignbit_b32 %0, %0, %1, %2")
OP2(bfi, "v_bfi_b32 %0, %0, %1, %2")
OP2(sub_f32, "v_sub_f32 %0, %0, %1")
OP2(mul_f32, "v_mul_f32 %0, %0, %1")
OP2(sad_u32, "v_sad_u32 %0, %0, %1, %2")
OP2(or3_b32, "v_or3_b32 %0, %0, %1, %2")
OP2(add3_u32, "v_add3_u32 %0, %0, %1, %2")
OP2(max_u32, "v_max_u32 %0, %0, %1")
OP2(sub_u32, "v_sub_u32 %0, %0, %1")

typedef void (*kern_t)(unsigned *, int);
struct Entry { const char *name; kern_t k; };

int main()
{
    unsigned *d;
    hipMalloc(&d, 256 * 2048 * sizeof(unsigned));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    Entry es[] = {{"v_min_f32", k_min_f32}, {"v_max_f32", k_max_f32}, {"v_min_i32", k_min_i32}, {"v_min_u32", k_min_u32},
                  {"v_add_f32", k_add_f32}, {"v_sub_f32", k_sub_f32}, {"v_mul_f32", k_mul_f32}, {"v_add_u32", k_add_u32},
                  {"v_fma_f32", k_fma_f32}, {"v_min3_f32", k_min3_f32}, {"v_med3_f32", k_med3_f32}, {"v_min3_i32", k_min3_i32},
                  {"v_med3_i32", k_med3_i32}, {"v_cndmask", k_cndmask}, {"v_cmp_lt_f32", k_cmp_lt_f32}, {"v_cmp_lt_u32", k_cmp_lt_u32},
                  {"v_pk_min_u16", k_pk_min_u16}, {"v_pk_min_f16", k_pk_min_f16}, {"v_xor_b32", k_xor_b32}, {"v_mov_b32", k_mov_b32},
                  {"v_perm_b32", k_perm_b32}, {"v_alignbit", k_alignbit}, {"v_bfi_b32", k_bfi}, {"v_sad_u32", k_sad_u32}, {"v_or3_b32", k_or3_b32}, {"v_add3_u32", k_add3_u32}, {"v_max_u32", k_max_u32}, {"v_sub_u32", k_sub_u32}};
    const int iters = 500;
    for (int wps : {1, 2, 3, 4, 8}) {
        printf("waves/SIMD=%d:", wps);
        for (auto &e : es) {
            const int grid = 256 * wps;
            e.k<<<grid, 256>>>(d, 5);
            hipDeviceSynchronize();
            hipEventRecord(a);
            e.k<<<grid, 256>>>(d, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            const double wave_instr = (double)grid * 4 * iters * 64.0;
            printf(" %s=%.2f", e.name, wave_instr / ms / 1e9);
        }
        printf("   [T wave-instr/s, chip]\n");
    }
    return 0;
}
