// stack_fast_mlz.hip -- dispatch of the LDS-column sigma / winsor kernels (stack_fast_mlz_impl.hpp) over the
// frame-count classes: NTOP = frames rounded up to a multiple of 16, 2 lanes per pixel up to 256 frames, else 4
#include <string>

#include "stack_kernels.h"

namespace nl {

bool launch_mlz_part_a(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream);
bool launch_mlz_part_b(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream);
bool launch_mlz_part_c(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream);
bool launch_mlz_part_d(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream);

int fast_mlz_supported(int mode, bool weighted, int n_frames)
{
    if (weighted || (mode != NL_ST_SIGMA && mode != NL_ST_WINSOR_SIGMA)) return 0;
    return (n_frames > 128 && n_frames <= 512) ? 1 : 0;
}

// (MlzSplit<MlzLayout<4, false, 512>>::N: low column 24, high column 32, scalars 8, median window 24)
int mlz_split_rows(int mode, int n_frames)
{
    return (mode == NL_ST_SIGMA && n_frames > 496 && n_frames <= 512) ? 88 : 0;
}

int decide_ml_supported(int mode, int n_frames, int64_t npix)
{
    return (fast_mlz_supported(mode, false, n_frames) && npix < kFastMaxPixels) ? 1 : 0;
}

// the zonal launch over the whole tile (the generic pass over its hand-over list is stack_fast_mlg.hip)
hipError_t launch_stack_sigma_mlz(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name,
                                  bool winsor)
{
    FastArgs f = fargs;
    f.in_list = nullptr;
    f.in_count = nullptr;
    f.in_capacity = 0;
    const int ntop = (args.n_frames + 15) / 16 * 16;
    const int lpp = args.n_frames <= 256 ? 2 : 4;
    // kernel names as rocprofv3 prints them (template arguments: LPP, WINSOR, NTOP, PHASE -- 0: the whole pass in one kernel;
    // the experimental passes of the selected class report their sorting kernel, 1, or the persistent kernel, 3)
    static const std::string *names = [] {
        static std::string t[3][2][33];
        for (int ph = 0; ph < 3; ph++)
            for (int w = 0; w < 2; w++)
                for (int c = 9; c <= 32; c++)
                    t[ph][w][c] = "stack_sigma_mlz_kernel<" + std::to_string(c <= 16 ? 2 : 4) + (w ? ", true, " : ", false, ") +
                                  std::to_string(16 * c) + (ph == 0 ? ", 0>" : (ph == 1 ? ", 1>" : ", 3>"));
        return &t[0][0][0];
    }();
    const bool selected = !winsor && mlz_split_rows(NL_ST_SIGMA, args.n_frames) != 0;
    const int ph = !selected ? 0 : (fargs.cols ? 1 : (fargs.persistent ? 2 : 0));
    *name = names[(ph * 2 + (winsor ? 1 : 0)) * 33 + ntop / 16].c_str();
    (void)lpp;
    const bool ok = launch_mlz_part_a(ntop, winsor, args, f, stream) || launch_mlz_part_b(ntop, winsor, args, f, stream) ||
                    launch_mlz_part_c(ntop, winsor, args, f, stream) || launch_mlz_part_d(ntop, winsor, args, f, stream);
    if (!ok) return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace nl
