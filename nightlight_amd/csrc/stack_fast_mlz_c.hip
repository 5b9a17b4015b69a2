// stack_fast_mlz_c.hip -- the LDS-column sigma / winsor kernels (stack_fast_mlz_impl.hpp) of the frame-count
// classes 368 .. 432 (4 lanes per pixel); the classes are spread over four files so that they build in parallel
#include "stack_fast_mlz_impl.hpp"

namespace nl {

bool launch_mlz_part_c(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream)
{
    return launch_mlz_classes<4>(ntop, winsor, args, f, stream, std::integer_sequence<int, 368, 384, 400, 416, 432>{});
}

}  // namespace nl
