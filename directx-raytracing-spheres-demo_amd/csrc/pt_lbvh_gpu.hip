// pt_lbvh_gpu.hip -- device LBVH builder.  (placeholder: the host builder is used until this lands)
#include "pt_lbvh_gpu.h"

namespace pt {

bool lbvh_gpu_available() { return false; }
LbvhGpu* lbvh_gpu_create() { return nullptr; }
void lbvh_gpu_destroy(LbvhGpu*) {}
hipError_t lbvh_gpu_build(LbvhGpu*, const float4*, uint32_t, PtBvhNode*, float4*, uint32_t*, hipStream_t, LbvhGpuInfo*) { return hipErrorNotSupported; }

}  // namespace pt
