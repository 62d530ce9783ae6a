// pt_lbvh_gpu.h -- device LBVH builder (Morton -> radix sort -> Karras hierarchy -> bottom-up refit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt_api.h"

namespace pt {

struct LbvhGpu;  // opaque: owns the builder's scratch buffers

struct LbvhGpuInfo {
    uint32_t depth;
    float pad;
    float bounds_min[3], bounds_max[3];
    float build_ms;
};

// false until the device builder is linked in (then pt_build_accel uses it unless PT_FLAG_HOST_LBVH)
bool lbvh_gpu_available();
LbvhGpu* lbvh_gpu_create();
void lbvh_gpu_destroy(LbvhGpu* b);

// d_sph: n spheres in original order (float4 {cx,cy,cz,r}).  Outputs (device): nodes[n-1], sorted[n], sorted_id[n].
// Synchronises the stream (the depth / bounds are read back).
hipError_t lbvh_gpu_build(LbvhGpu* b, const float4* d_sph, uint32_t n, PtBvhNode* d_nodes, float4* d_sorted, uint32_t* d_sorted_id,
                          hipStream_t stream, LbvhGpuInfo* info);

// Adopt a topology built on the host (links + leaf order; the boxes are computed here by the refit passes).
hipError_t lbvh_gpu_adopt(LbvhGpu* b, const float4* d_sph, uint32_t n, const PtBvhNode* h_nodes, const uint32_t* h_sorted_id, uint32_t depth,
                          PtBvhNode* d_nodes, float4* d_sorted, uint32_t* d_sorted_id, hipStream_t stream, LbvhGpuInfo* info);

// Refit after the spheres moved (same count, same topology as the last build): re-gather the Morton-ordered sphere
// copy and recompute every box bottom-up.  Asynchronous on `stream`; flags (n words) and hdr (16 words) are scratch the
// caller owns (one set per stream that may refit concurrently).
hipError_t lbvh_gpu_refit(LbvhGpu* b, const float4* d_sph, uint32_t n, PtBvhNode* d_nodes, float4* d_sorted, const uint32_t* d_sorted_id,
                          uint32_t* d_flags, uint32_t* d_hdr, uint32_t depth, hipStream_t stream);

// Small scenes (lbvh_gpu_refit_fused_possible): upload + bounds + gather + refit in ONE launch.  src: the new spheres as the device sees
// them (device memory, or pinned host memory); d_sph: the device array they are copied to.
bool lbvh_gpu_refit_fused_possible(uint32_t n);
hipError_t lbvh_gpu_refit_fused(LbvhGpu* b, const float4* src, float4* d_sph, uint32_t n, PtBvhNode* d_nodes, float4* d_sorted, const uint32_t* d_sorted_id,
                                uint32_t* d_hdr, hipStream_t stream);

// 4-wide, quantised view of a finished binary tree (n_nodes internal nodes): d_wide = n_nodes * 4 float4 (64-byte records at the
// binary node's index; only even-depth nodes are written / reachable).  Asynchronous on `stream`.
hipError_t lbvh_gpu_collapse4(const PtBvhNode* d_nodes, uint32_t n_nodes, float4* d_wide, hipStream_t stream);

}  // namespace pt
