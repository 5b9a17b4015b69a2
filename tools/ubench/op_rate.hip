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
OP2(max_i32, "v_max_i32 %0, %0, %1")
OP2(pk_min_i16, "v_pk_min_i16 %0, %0, %1")
OP2(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
OP2(min_f16, "v_min_f16 %0, %0, %1")
OP2(min_u16, "v_min_u16 %0, %0, %1")
OP2(min_f32_dpp, "v_min_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(min_f32_dpp_neg, "v_min_f32_dpp %0, -%1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(add_f32_dpp, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(minimum3_f32, "v_minimum3_f32 %0, %0, %1, %2")
OP2(maximum3_f32, "v_maximum3_f32 %0, %0, %1, %2")
OP2(addc_u32, "v_addc_co_u32 %0, vcc, 0, %0, vcc")
OP2(add_f32_abs, "v_add_f32_e64 %0, %0, |%1|")
OP2(sub_f32_e64neg, "v_sub_f32_e64 %0, %0, -%1")
OP2(fmac_f32, "v_fmac_f32 %0, %1, %2")
OP2(bfe_u32, "v_bfe_u32 %0, %0, 3, 1")
OP2(lshrrev, "v_lshrrev_b32 %0, 1, %0")
OP2(and_b32, "v_and_b32 %0, %0, %1")
OP2(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
OP2(and_or, "v_and_or_b32 %0, %0, %1, %2")
OP2(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
OP2(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP2(mul_f32_e64, "v_mul_f32_e64 %0, %0, %1 mul:2")
OP2(max_f32_abs, "v_max_f32_e64 %0, %0, |%1|")
OP2(rcp_f32, "v_rcp_f32 %0, %0")
OP2(sqrt_f32, "v_sqrt_f32 %0, %0")
OP2(cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
OP2(fmamk, "v_fmamk_f32 %0, %0, 0x40400000, %1")
OP2(fmaak, "v_fmaak_f32 %0, %0, %1, 0x40400000")

#define OP64(name, asmstr)                                                                   \
    __global__ __launch_bounds__(256) void k_##name(unsigned *out, int iters)                  \
    {                                                                                        \
        double r[16];                                                                        \
        for (int i = 0; i < 16; i++) r[i] = __hiloint2double(threadIdx.x * 7 + i, 0x3f800000 + i);  \
        double a = __hiloint2double(0x3f800001, 0x3f800002), b = __hiloint2double(0x3f000001, 0x3f000002); \
        asm volatile("" : "+v"(a), "+v"(b));                                                 \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int rep = 0; rep < 4; rep++) {                             \
                _Pragma("unroll") for (int i = 0; i < 16; i++)                                \
                    asm volatile(asmstr : "+v"(r[i]) : "v"(a), "v"(b));                      \
            }                                                                                \
        }                                                                                    \
        unsigned s = 0;                                                                      \
        for (int i = 0; i < 16; i++) s += __double2loint(r[i]) + __double2hiint(r[i]);       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
    }
OP64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
OP64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
OP64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
OP64(pk_add_f32_neg, "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]")
OP64(pk_mov_b32, "v_pk_mov_b32 %0, %1, %2")
OP64(min_f64, "v_min_f64 %0, %0, %1")
OP64(add_f64, "v_add_f64 %0, %0, %1")

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
                  {"v_perm_b32", k_perm_b32}, {"v_alignbit", k_alignbit}, {"v_bfi_b32", k_bfi}, {"v_sad_u32", k_sad_u32}, {"v_or3_b32", k_or3_b32}, {"v_add3_u32", k_add3_u32}, {"v_max_u32", k_max_u32}, {"v_sub_u32", k_sub_u32}, {"max_i32", k_max_i32}, {"pk_min_i16", k_pk_min_i16}, {"pk_max_i16", k_pk_max_i16}, {"min_f16", k_min_f16}, {"min_u16", k_min_u16}, {"min_f32_dpp", k_min_f32_dpp}, {"min_f32_dpp_neg", k_min_f32_dpp_neg}, {"mov_dpp", k_mov_dpp}, {"add_f32_dpp", k_add_f32_dpp}, {"minimum3_f32", k_minimum3_f32}, {"maximum3_f32", k_maximum3_f32}, {"addc_u32", k_addc_u32}, {"add_f32_abs", k_add_f32_abs}, {"sub_f32_e64neg", k_sub_f32_e64neg}, {"fmac_f32", k_fmac_f32}, {"bfe_u32", k_bfe_u32}, {"lshrrev", k_lshrrev}, {"and_b32", k_and_b32}, {"lshl_add", k_lshl_add}, {"and_or", k_and_or}, {"cvt_f32_u32", k_cvt_f32_u32}, {"mad_u32_u24", k_mad_u32_u24}, {"mul_f32_e64", k_mul_f32_e64}, {"max_f32_abs", k_max_f32_abs}, {"rcp_f32", k_rcp_f32}, {"sqrt_f32", k_sqrt_f32}, {"cmp_cnd", k_cmp_cnd}, {"fmamk", k_fmamk}, {"fmaak", k_fmaak}, {"pk_add_f32", k_pk_add_f32}, {"pk_mul_f32", k_pk_mul_f32}, {"pk_fma_f32", k_pk_fma_f32}, {"pk_add_f32_neg", k_pk_add_f32_neg}, {"pk_mov_b32", k_pk_mov_b32}, {"min_f64", k_min_f64}, {"add_f64", k_add_f64}};
    const int iters = 500;
    for (int wps : {1, 3, 8}) {
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
            printf(" %s=%.3f\n", e.name, wave_instr / ms / 1e9);
        }
        printf("   [T wave-instr/s, chip]\n");
    }
    return 0;
}
