// pt_kernels.h -- launch wrappers of the kernel set (pt_kernels.hip), used by the C-ABI implementation.
#pragma once

#include "pt_device.h"
#include "pt_post.h"

namespace pt {

constexpr uint32_t kTraverseThreads = 512;  // launch bound; 8 waves share one LDS copy of the BVH
// threads per traverse-type workgroup: 512 when the BVH is staged in LDS (amortises the copy), 256 when it is read
// from global memory (the per-lane stacks are then the only LDS use, and smaller groups keep more waves resident)
inline uint32_t traverse_threads(bool lds_scene) { return lds_scene ? kTraverseThreads : 256u; }
constexpr uint32_t kShadeThreads = 256;
constexpr uint32_t kTailThreads = 256;
constexpr uint32_t kFusedThreads = 512;  // launch bound of the fused trace+shade kernel (launched with 512 or 256 threads)

// dynamic LDS a traverse-type launch needs (scene copy if lds_scene, plus the per-lane stacks)
uint32_t traverse_lds_bytes_for(uint32_t n_nodes, uint32_t n, uint32_t depth, bool lds_scene, uint32_t threads = 0);

hipError_t launch_primary(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& q, const Scratch& scratch,
                          float4* out, const FrameCounters& fc, uint32_t grid, hipStream_t stream);
// primary beams: per-8x8-block candidate sphere lists for the primary pass (lists: pm.n_slots / 64 records of 16 dwords)
hipError_t launch_beams(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, float slack, uint32_t* lists, hipStream_t stream);
hipError_t launch_traverse(const SceneView& sv, const RayQueue& q, const uint32_t* count_ptr, uint32_t grid, hipStream_t stream);
hipError_t launch_traverse_dyn(const SceneView& sv, const RayQueue& q, const uint32_t* count_ptr, uint32_t* cursor, unsigned long long* totals, uint32_t grid,
                               hipStream_t stream);
hipError_t launch_shade(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const RayQueue& qout,
                        const Scratch& scratch, float4* out, const uint32_t* count_in, uint32_t* count_out, uint32_t grid, hipStream_t stream);
hipError_t launch_trace(const SceneView& sv, const float* o, const float* d, uint32_t n_rays, float tmin, int use_bvh, float* out_t,
                        uint32_t* out_id, uint2* out_visits, hipStream_t stream);
hipError_t launch_tail(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const Scratch& scratch, float4* out,
                       const uint32_t* count_ptr, unsigned long long* tail_rays, unsigned long long* totals, uint32_t grid, hipStream_t stream);
hipError_t launch_flush_counters(uint32_t* counts, uint32_t n_counts, unsigned long long* tail, unsigned long long* totals, uint32_t* host_counts,
                                 hipStream_t stream);
hipError_t launch_bounce(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const RayQueue& qout,
                         const Scratch& scratch, float4* out, const uint32_t* count_in, uint32_t* count_out, const FrameCounters& fc,
                         bool primary, bool loop, bool inline2, uint32_t threads, uint32_t grid, hipStream_t stream);
hipError_t launch_unpack_tiles(const float4* packed, float4* frame, uint32_t w, uint32_t h, uint32_t ts, uint32_t tiles_x, uint32_t first0,
                               uint32_t run, uint32_t stride, uint32_t n_parts, uint64_t part_stride, bool rgb, hipStream_t stream);
hipError_t launch_pack_rgb(const float4* src, float* dst, uint64_t n, hipStream_t stream);
// bit 31 of the AlphaMode word of every device material := the sphere has texture maps (tex_maps null: none has)
hipError_t launch_material_map_flags(float4* mats, const uint32_t* tex_maps, uint32_t n, hipStream_t stream);
// alpha-tested hits: out[k] = sorted_id[k] | alpha_class[sorted_id[k]] << 30 (the leaf ids of SceneView::sorted_id)
hipError_t launch_leaf_ids(const uint32_t* sorted_id, const uint32_t* alpha_class, uint32_t n, uint32_t* out, hipStream_t stream);

hipError_t launch_di(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, float4* di, uint2* primary_hit, unsigned long long* ray_counter, uint32_t grid,
                     hipStream_t stream);
hipError_t launch_tonemap(const float4* hdr, uint32_t* out, uint32_t n, const PtToneMapParams& p, hipStream_t stream);
hipError_t launch_accumulate(float4* accum, const float4* rad, uint32_t n, uint32_t frames_accumulated, hipStream_t stream);

}  // namespace pt
