// pt_bounce_lds16.hip -- the bounce_kernel instances (pt_trace.h) for kLds = true, StackT = uint16_t: a translation unit of its own so
// that the instances of the fused trace + shade kernel compile in parallel.
#include "pt_trace.h"

namespace pt {

hipError_t launch_bounce_lds16(PT_BOUNCE_LAUNCHER_ARGS)
{
    return launch_bounce_for<true, uint16_t>(sv, pm, fp, qin, qout, scratch, out, count_in, count_out, fc, primary, loop, inline2, threads, grid, stream);
}

}  // namespace pt
