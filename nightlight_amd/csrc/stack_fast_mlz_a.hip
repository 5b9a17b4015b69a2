// stack_fast_mlz_a.hip -- the LDS-column sigma / winsor kernels (stack_fast_mlz_impl.hpp) of the frame-count
// classes 144 .. 256 (2 lanes per pixel); the classes are spread over four files so that they build in parallel
#include "stack_fast_mlz_impl.hpp"

namespace nl {

bool launch_mlz_part_a(int ntop, bool winsor, const StackArgs &args, const FastArgs &f, hipStream_t stream)
{
    return launch_mlz_classes<2>(ntop, winsor, args, f, stream, std::integer_sequence<int, 144, 160, 176, 192, 208, 224, 240, 256>{});
}

}  // namespace nl
