// pt_lbvh.h -- LBVH over analytic spheres (replaces the DXR BLAS/TLAS of Source/Scene.ixx:225-284).
// Morton codes (30 bit) of sphere centres -> sort -> Karras 2012 radix-tree hierarchy -> bottom-up AABBs.
// Host builder (reference implementation, also the PT_FLAG_HOST_LBVH path); the device builder in
// pt_lbvh_gpu.hip produces the identical tree.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/pt_api.h"

namespace pt {

struct LbvhResult {
    std::vector<PtBvhNode> nodes;      // n - 1 internal nodes, node 0 is the root (empty when n == 1)
    std::vector<PtSphere> sorted;      // spheres in Morton order
    std::vector<uint32_t> sorted_id;   // Morton order -> original id
    uint32_t depth = 0;                // max number of internal nodes on a root-to-leaf path
    float bounds_min[3]{}, bounds_max[3]{};
    float pad = 0;                     // AABB padding applied to every leaf box
};

// 30-bit Morton code of a point normalised to [0,1]^3 (10 bits per axis, x most significant of each triple)
uint32_t morton30(float x, float y, float z);

// AABB padding that makes the slab test conservative w.r.t. intersect_sphere (DESIGN.md "LBVH"):
// 2^-17 * max |coordinate| over the scene bounds.
float lbvh_padding(const float bmin[3], const float bmax[3]);

void build_lbvh_host(const PtSphere* spheres, uint32_t n, LbvhResult& out);
// SAH topology for small scenes (same record format; see pt_lbvh.cpp)
void build_sah_host(const PtSphere* spheres, uint32_t n, LbvhResult& out);

}  // namespace pt
