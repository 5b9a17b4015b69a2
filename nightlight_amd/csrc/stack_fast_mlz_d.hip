// stack_fast_mlz_d.hip -- the LDS-column sigma / winsor kernels (stack_fast_mlz_impl.hpp) of the frame-count
// classes 448 .. 512 (4 lanes per pixel); the classes are spread over four files so that they build in parallel
#include "stack_fast_mlz_impl.hpp"

namespace nl {

static_assert(MlzSplit<MlzLayout<4, false, 512>>::N == 88, "mlz_split_rows() in stack_fast_mlz.hip");

bool launch_mlz_part_d(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream)
{
    return launch_mlz_classes<4>(ntop, winsor, args, f, stream, std::integer_sequence<int, 448, 464, 480, 496, 512>{});
}

}  // namespace nl
