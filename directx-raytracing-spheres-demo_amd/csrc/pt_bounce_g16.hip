// pt_bounce_g16.hip -- the bounce_kernel instances (pt_trace.h) for kLds = false, StackT = uint16_t: a translation unit of its own so
// that the instances of the fused trace + shade kernel compile in parallel.
#include "pt_trace.h"

namespace pt {

hipError_t launch_bounce_g16(PT_BOUNCE_LAUNCHER_ARGS)
{
    return launch_bounce_for<false, uint16_t>(sv, pm, fp, qin, qout, scratch, out, count_in, count_out, fc, primary, loop, inline2, threads, grid, stream);
}

}  // namespace pt
