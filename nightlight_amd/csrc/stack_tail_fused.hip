// stack_tail_fused.hip -- the tail of a short-listed sigma pass as ONE launch.
//
// Behind the dominant kernel of a 65 ... 128-frame sigma clip come the generic pass over the pixels it handed over
// (stack_fast_mlg.hip: a few dozen waves, 33 - 37 us = the latency of one of them) and the bit-exact replay of its undecidable
// pixels (stack_exact_coop.hip: one wave per pixel, 32 us = the latency of one pixel).  Both only depend on the dominant kernel.
// Run one after the other they cost two latencies; run side by side on two streams (the protocol of nlstack_api.hip since
// round 2) they cost one latency plus the cross-stream join in front of the next kernel: 14 us, and 12 us that the dominant kernel
// itself runs longer behind it (kernel timelines) -- a tenth of a 512-row tile's pass.  HIP offers no
// other way for two kernels of one stream to overlap on gfx9 (hipExtAnyOrderLaunch is ignored there: measured).
//
// Here the two kernels are the upper and the lower workgroups of one grid: the last gen_blocks workgroups run mlg_body, the
// others coop_body on the exact list as the dominant kernel left it (list part 0; the snapshot protocol of fast_common.hpp does not
// care who looks first).  The replay of what the generic pass adds to the list follows as before.  Every workgroup claims the
// generic pass's 48 KiB of LDS, so three replay workgroups fit a CU: dispatched only while the exact list is short
// (nlstack_api.hip: kTailFusedMaxList), which is where the join weighs most.
#undef NL_ROUND_STATS
#undef NL_PROBE
#define NL_TAIL_FUSED_TU
#include "stack_fast_mlg.hip"
#include "stack_exact_coop.hip"

namespace nl {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
void stack_sigma_tail_kernel(StackArgs pg, FastArgs qg, StackArgs pe, unsigned gen_blocks)
{
    extern __shared__ float replay_columns[];
    // (the replay's workgroups first: the generic pass's waves are the long pole -- behind the replay's they start a
    // microsecond later; in front of them the replay's last workgroups started, and the grid ended, 5 us later)
    const unsigned replay_blocks = gridDim.x - gen_blocks;
    if (blockIdx.x < replay_blocks) coop_body<false, false, 1, 2>(pe, replay_columns, blockIdx.x, replay_blocks);
    else mlg_body<1, false>(pg, qg, blockIdx.x - replay_blocks, gen_blocks);
}

int tail_fused_supported(int mode, bool weighted, int n_frames)
{
    return (mode == NL_ST_SIGMA && !weighted && n_frames > 64 && n_frames <= kMlNS) ? 1 : 0;
}

hipError_t launch_stack_sigma_tail(const StackArgs &generic, const FastArgs &fargs, unsigned gen_blocks,
                                   const StackArgs &replay, unsigned replay_blocks, hipStream_t stream)
{
    // LDS of the replay part: samples + 2 scratch columns of 16 bits (coop_columns(NL_ST_SIGMA, false) in stack_exact_coop.hip)
    const size_t lds = (size_t)replay.n_frames * sizeof(float) * 2;
    hipLaunchKernelGGL(stack_sigma_tail_kernel, dim3(gen_blocks + replay_blocks), dim3(64), lds, stream, generic, fargs, replay,
                       gen_blocks);
    return hipGetLastError();
}

}  // namespace nl
