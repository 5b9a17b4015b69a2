// stack_fast_decide.hip -- weighted sigma / winsorized sigma clipping: the DECISION pass.
//
// StackSigmaWeighted / StackWinsorSigmaWeighted (stack.go:442-531, 710-829) reject exactly as their unweighted
// twins -- median and standard deviation ignore the weights -- and differ only in the result: the weighted mean of
// the survivors IN THE ORDER the quickselects and the clip swaps left them, with weights that followed only the
// clip swaps (stack.go:487).  That order has to be replayed (stack_exact_coop*.hip), but the bounds of every
// clipping round need not be: the register-resident kernel (stack_fast_sigma_impl.hpp, RECORD) decides them with its
// interval guard -- on the sorted column, no permutation involved -- and leaves thresholds per round and pixel; the
// replay then only permutes and clips, without the sequential sums of MeanStdDev (stats.go:246-261) and without the
// winsorization loop (stack.go:646-672: about 20 such sums per clipping round).  Pixels the guard cannot decide
// (1e-4 .. 1e-2 of them), the NaN borders and pixels that clip more than the zones hold are replayed in full.
#include <string.h>

#include <string>

#define NL_STAT(i, x) ((void)0)
#include "stack_fast_sigma_impl.hpp"

namespace nl {

int decide_supported(int mode, int n_frames, int64_t npix)
{
    if (mode != NL_ST_SIGMA && mode != NL_ST_WINSOR_SIGMA) return 0;
    return (n_frames >= 33 && n_frames <= 128 && npix < kFastMaxPixels) ? 1 : 0;
}

template <int NS, bool WINSOR>
static void launch_decide(const StackArgs &args, unsigned blocks, hipStream_t stream, const char **name)
{
    static const std::string names[2] = {
        std::string("stack_sigma_fast_kernel<") + std::to_string(NS) + ", true, " + (WINSOR ? "true" : "false") + ", false, true, false>",
        std::string("stack_sigma_fast_kernel<") + std::to_string(NS) + ", true, " + (WINSOR ? "true" : "false") + ", true, true, false>"};
    FastArgs f;
    memset(&f, 0, sizeof f);
    if (args.n_frames == NS) {
        *name = names[1].c_str();
        hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, true, WINSOR, true, true>), dim3(blocks), dim3(256), 0, stream, args, f);
    } else {
        *name = names[0].c_str();
        hipLaunchKernelGGL((stack_sigma_fast_kernel<NS, true, WINSOR, false, true>), dim3(blocks), dim3(256), 0, stream, args, f);
    }
}

template <bool WINSOR>
static void launch_decide_sized(const StackArgs &args, hipStream_t stream, const char **name)
{
    const unsigned blocks = (unsigned)((args.npix + 255) / 256);
    const int n = args.n_frames;
    if (n <= 48)       launch_decide<48, WINSOR>(args, blocks, stream, name);
    else if (n <= 64)  launch_decide<64, WINSOR>(args, blocks, stream, name);
    else if (n <= 80)  launch_decide<80, WINSOR>(args, blocks, stream, name);
    else if (n <= 96)  launch_decide<96, WINSOR>(args, blocks, stream, name);
    else if (n <= 112) launch_decide<112, WINSOR>(args, blocks, stream, name);
    else               launch_decide<128, WINSOR>(args, blocks, stream, name);
}

hipError_t launch_stack_sigma_decide(const StackArgs &args, hipStream_t stream, bool winsor, const char **name)
{
    if (winsor) launch_decide_sized<true>(args, stream, name);
    else        launch_decide_sized<false>(args, stream, name);
    return hipGetLastError();
}

}  // namespace nl
