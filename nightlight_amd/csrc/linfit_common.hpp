// linfit_common.hpp -- helpers shared by the linear-fit kernels: stack_linfit.hip (bit-exact cascade) and
// stack_linfit_guard.hip (guarded stages in front of it).
#pragma once
#include "fast_ml_common.hpp"

namespace nl {

__device__ __forceinline__ float sqrt_go(float x)      // float32(math.Sqrt(float64(x))), stats.go:259
{
    return (float)__builtin_sqrt((double)x);
}

// These make the compiler forget what it knows about a value (no instruction is
// emitted).  Used between the passes of an iteration: otherwise it keeps the
// 128 per-sample liveness factors and the 128 differences x-ymean of one pass
// in registers for the next pass instead of recomputing them (2-3x the VGPRs).
__device__ __forceinline__ float opaque_f(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ unsigned opaque_u(unsigned x) { asm volatile("" : "+v"(x)); return x; }
template <int NW>
__device__ __forceinline__ void forget_words(unsigned (&w)[NW])
{
    static_assert(NW <= 4, "at most 128 samples");
    asm volatile("" : "+v"(w[0]));
    if constexpr (NW > 1) asm volatile("" : "+v"(w[1]));
    if constexpr (NW > 2) asm volatile("" : "+v"(w[2]));
    if constexpr (NW > 3) asm volatile("" : "+v"(w[3]));
}

// the chunk classes are wave-uniform: a volatile asm in each arm keeps the compiler from
// if-converting the scalar branches (it would execute both arms and select)
#define NL_KEEP_BRANCH asm volatile("")

__device__ __forceinline__ float max3_asm(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3_asm(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 1 if x < 0 (sign bit), else 0 -- integer arithmetic on purpose: a compare
// would produce a lane mask in SGPRs per element
__device__ __forceinline__ unsigned sign_bit(float x) { return (unsigned)__float_as_int(x) >> 31; }

// CONT = false: first stage, the grid covers the tile.  CONT = true: continuation
// stage, grid-stride over the pixels the previous stage handed over; their
// liveness masks come from memory (the sorted column is re-created: sorting is
// deterministic, so the mask positions still mean the same samples).
struct LinfitStage {
    const unsigned *in_list;  const unsigned *in_count;  const uint4 *in_state;  unsigned in_capacity;
    unsigned *out_list;       unsigned *out_count;       uint4 *out_state;       unsigned out_capacity;
    int max_iters;            // fit iterations this stage may run per pixel (0: unlimited)
};

}  // namespace nl
