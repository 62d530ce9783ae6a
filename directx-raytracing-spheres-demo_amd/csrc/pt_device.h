// pt_device.h -- device-side data layout shared by the kernels (pt_kernels.hip) and the C-ABI
// implementation (pt_api.hip).  See DESIGN.md "Data layout in HBM".
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_bsdf.h"
#include "pt_texture.h"
#include "pt_light.h"

namespace pt {

// ---- ray queue: SoA of three float4 streams (48 B per ray) + one 8-B hit stream --------------------
//   q0[i] = { o.x, o.y, o.z, bits(slot) }        slot = path slot (-> pixel, scratch index)
//   q1[i] = { d.x, d.y, d.z, bits(rng state) }
//   q2[i] = { T.r, T.g, T.b, bits(flags) }       flags = bounce | sample << 8 | srad_dirty << 24
//   hit[i] = { bits(t), sphere id (0xFFFFFFFF = miss) }
struct RayQueue {
    float4* q0;
    float4* q1;
    float4* q2;
    uint2* hit;
};

constexpr uint32_t kFlagBounceMask = 0xFFu;
constexpr uint32_t kFlagSampleShift = 8;
constexpr uint32_t kFlagSampleMask = 0xFFFFu;
constexpr uint32_t kFlagDirty = 1u << 24;
constexpr uint32_t kFlagViaTransmission = 1u << 25;  // row N4: this sample left the primary surface through the transmission lobe
constexpr uint32_t kMissId = 0xFFFFFFFFu;
constexpr uint32_t kMaterialHasMaps = 0x80000000u;  // device copy of PtMaterial::AlphaMode, bit 31: the sphere has texture maps (pt_set_textures)

// Alpha-tested hits (DESIGN.md spec S10; Scene.ixx:242-243, RaytracingHelpers.hlsli:19-43, ShadingHelpers.hlsli:105-115).  A sphere whose
// AlphaMode is not Opaque is classified on the host when materials or texture maps change:
//   kAlphaVisible    every crossing is accepted: Opaque, or constant alpha (no base-colour map is sampled) >= AlphaCutoff
//   kAlphaInvisible  no crossing is accepted: constant alpha < AlphaCutoff (or NaN) -- the sphere does not exist for rays
//   kAlphaTested     alpha = BaseColor.a * the base-colour map's alpha at the crossing: tested per crossing, near root then far root
// The class rides in the two top bits of the leaf ids the traversal reads (SceneView::sorted_id; sphere ids stay below 2^30), so
// a scene without such spheres pays one compare per successful sphere test and nothing else.
enum : uint32_t { kAlphaVisible = 0u, kAlphaTested = 1u, kAlphaInvisible = 2u };
constexpr uint32_t kIdMask = 0x3FFFFFFFu;
constexpr uint32_t kIdClassShift = 30;

// ---- scene view ------------------------------------------------------------------------------------
// BVH node = 64 B = 4 float4 (PtBvhNode of include/pt_api.h):
//   n0 = lo0.xyz, hi0.x   n1 = hi0.yz, lo1.xy   n2 = lo1.z, hi1.xyz   n3 = child0, child1, parent, pad (ints)
struct SceneView {
    const float4* nodes;        // n_nodes * 4
    const float4* wide;         // 4-wide quantised view of the tree (n_nodes * 4, pt_lbvh_gpu.hip collapse4_kernel) or null; global-memory scenes
    const float4* sph_sorted;   // Morton order {cx,cy,cz,r}
    const uint32_t* sorted_id;  // Morton order -> original sphere id | alpha class << 30 (kIdMask, kIdClassShift)
    const float4* sph;          // original order (shade)
    const float4* mats;         // original order, PtMaterial as 4 float4
    uint32_t n;                 // spheres
    uint32_t n_nodes;           // internal nodes (n - 1; 0 when n == 1)
    uint32_t stack_depth;       // traversal stack entries per lane
    uint32_t descent_cap;       // global-memory scenes: node visits per lane before the wave turns to its sphere tests (0 = unbounded)
    uint32_t lds_scene;         // 1: kernels stage nodes + sph_sorted + sorted_id in LDS
    float env[4];               // SceneData.EnvironmentLightColor
    // row N1 (textured spheres); null when the scene has no textures
    const TexView* tex;         // texture table
    const uint32_t* tex_maps;   // per sphere: 7 texture indices (TextureMapType order) + 1 "has any" flag
    const float4* rot;          // per sphere: object rotation quaternion (x, y, z, w)
    uint32_t env_tex;           // environment map: index into tex, kNoTexture = EnvironmentLightColor / sky
    uint32_t env_cube;          // 1: tex[env_tex .. env_tex + 5] are the faces of a cube map; 0: tex[env_tex] is a lat-long map
    float env_xf[9];            // upper 3x3 of SceneData.EnvironmentLightTransform, row-major
    // row N4 (sphere-light direct illumination): ids of the emissive spheres, in id order
    const uint32_t* lights;
    uint32_t n_lights;
    // alpha-tested hits: per sphere (original order) its class, or null when every sphere is kAlphaVisible; alpha_tested = some
    // sphere is kAlphaTested (only then do the kernels without a texture variant of their own need their kAlphaTex form)
    const uint32_t* alpha_class;
    uint32_t alpha_tested;
};

// ---- slot -> pixel mapping ---------------------------------------------------------------------------
// A slot is 64-aligned to an 8x8 pixel block so that one wave64 = one 8x8 block of pixels.
//   mode 0 (rect):  block b = slot / 64 over ceil(w/8) x ceil(h/8) blocks of the rect
//   mode 1 (tiles): tile k = slot / ts^2 is this rank's k-th tile = global tile (rank + k * world)
struct PixelMap {
    uint32_t mode;
    uint32_t img_w, img_h;        // RenderSize
    uint32_t rx, ry, rw, rh;      // rect (mode 0)
    uint32_t blocks_x;            // ceil(rw / 8) (mode 0)
    uint32_t ts, tiles_x, tiles_total;  // mode 1 (ts = 1 << ts_shift)
    // mode 1 ownership: the tiles t with first <= t % stride < first + run, in increasing t (pt_set_partition_ex);
    // the plain rank/world interleave is (first, run, stride) = (rank, 1, world)
    uint32_t first, run, stride;
    uint32_t ts_shift;
    uint32_t n_slots;
    float inv_blocks_x, inv_tiles_x, inv_run;  // reciprocals for fast_div
    uint32_t exact_div;               // 1: slot counts too large for the float-reciprocal division
};

struct PixelRef {
    uint32_t px, py;     // global pixel
    uint32_t out_index;  // index into the output float4 buffer
    bool valid;
};

// n / d for a launch-invariant divisor: float reciprocal + one exact correction step.  Valid while n < 2^22 (the float
// conversion is exact and the quotient estimate is off by at most one); PixelMap::exact_div selects the plain division
// for larger slot counts.  A 32-bit integer division costs ~25 VALU instructions on gfx950, this ~8.
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t d, float inv_d, bool exact_div)
{
    if (exact_div) return n / d;
    uint32_t q = (uint32_t)((float)n * inv_d);
    const int32_t r = (int32_t)(n - q * d);
    if (r < 0) q--; else if ((uint32_t)r >= d) q++;
    return q;
}

__device__ __forceinline__ PixelRef slot_to_pixel(const PixelMap& m, uint32_t slot)
{
    PixelRef r;
    if (m.mode == 0) {
        const uint32_t b = slot >> 6, l = slot & 63u;
        const uint32_t by = fast_div(b, m.blocks_x, m.inv_blocks_x, m.exact_div != 0), bx = b - by * m.blocks_x;
        const uint32_t x = bx * 8u + (l & 7u), y = by * 8u + (l >> 3);
        r.valid = x < m.rw && y < m.rh;
        r.px = m.rx + x;
        r.py = m.ry + y;
        r.out_index = y * m.rw + x;
    } else {
        // tile edge ts is a power of two >= 8 (ts_shift = log2 ts): shifts and masks only, except the tile -> (tx, ty) split
        const uint32_t ts2_shift = 2u * m.ts_shift;
        const uint32_t k = slot >> ts2_shift, w = slot & ((1u << ts2_shift) - 1u);
        const uint32_t bpt_shift = m.ts_shift - 3u;  // 8x8 blocks per tile row = ts / 8
        const uint32_t b = w >> 6, l = w & 63u;
        const uint32_t lx = ((b & ((1u << bpt_shift) - 1u)) << 3) + (l & 7u), ly = ((b >> bpt_shift) << 3) + (l >> 3);
        uint32_t gt;
        if (m.run == 1u) {
            gt = m.first + k * m.stride;
        } else {
            const uint32_t q = fast_div(k, m.run, m.inv_run, m.exact_div != 0);
            gt = m.first + q * m.stride + (k - q * m.run);
        }
        const uint32_t ty = fast_div(gt, m.tiles_x, m.inv_tiles_x, m.exact_div != 0), tx = gt - ty * m.tiles_x;
        r.px = (tx << m.ts_shift) + lx;
        r.py = (ty << m.ts_shift) + ly;
        r.valid = gt < m.tiles_total && r.px < m.img_w && r.py < m.img_h;
        r.out_index = (k << ts2_shift) + (ly << m.ts_shift) + lx;
    }
    return r;
}

// a share of a primary-beam list build carried by a primary pass (bounce_kernel): blocks [first_block, first_block + n_blocks) of `lists`,
// for a camera at `centre` (this frame's orientation) with Beam::slack `slack`; n_blocks = 0: none
struct BeamJob {
    uint32_t* lists;
    uint32_t first_block, n_blocks;
    float centre[3];
    float slack;
    float right[3], up[3], forward[3];  // the orientation the lists are made for (CameraParams::Right / Up / Forward)
    float margin_px;                     // make_beam: pixels added on every side of a block's outline
};

struct FrameParams {
    CameraParams cam;
    uint32_t frame_index, bounces, spp, rr_enabled;
    float throughput_threshold;
    float inv_spp;        // 1 / (float)spp
    uint32_t di_enabled;  // IsDIEnabled and the scene has emitters: Scratch::di holds this frame's estimate
    const uint32_t* beam_lists;  // primary beams: one 16-dword record per 64 slots {count, sphere ids}; null = every primary ray traverses
    BeamJob beam_job;
};

// Per-frame device counters, double buffered by frame parity so that the first kernel of a frame can append to this
// frame's counters while its block 0 folds the PREVIOUS frame's into the running totals and zeroes them for the next.
//   counts[k]  = size of ray queue k of this frame (counts[0] = all slots), appended with one atomicAdd per workgroup
//   tail_rays  = rays spawned inside looping kernels of this frame (they never enter a queue)
//   fold_*     = the other parity's counters (a finished frame)
//   totals[0]  = running sum of secondary rays over folded frames, totals[1] = secondary rays of the last folded frame
struct FrameCounters {
    uint32_t* counts;
    uint32_t* fold_counts;
    uint32_t n_counts;  // counts[0..n_counts] are valid in both arrays; each array is followed by n_counts + 1 work
                        // cursors (cursor k hands out the rays of queue k to the ray-replacement traverse kernel)
    unsigned long long* tail_rays;
    unsigned long long* fold_tail;
    unsigned long long* totals;
    uint32_t* host_counts;  // host-mapped (pinned) copy of the queue sizes of the frame being folded: launch-grid estimates
                            // for later frames reach the host without a copy call; may be null
    // Segmented hand-over from the primary pass to the looping pass (null = dense queue): workgroup b of the primary pass
    // appends its surviving rays to its own segment [b * seg_cap, ...) of the queue, through a counter in LDS -- no barrier
    // and no global atomic per batch -- and leaves the segment's size in seg_counts[b]; the looping pass maps a dense index
    // to (segment, offset) through a prefix sum it builds in LDS.
    uint32_t* seg_counts;
    uint32_t n_segs;        // == grid size of the primary pass, <= kMaxSegs
    uint32_t seg_cap;       // entries per segment
    uint32_t fuse_loop;     // 1: the primary pass finishes its own segment itself (no separate looping pass is launched)
};

constexpr uint32_t kMaxSegs = 2048;

// per-slot scratch (only touched when needed, see shade kernel)
struct Scratch {
    float4* sample_rad;   // sampleRadiance of the sample in flight (valid when kFlagDirty)
    float4* radiance;     // sum over finished samples (spp > 1)
    uint2* primary_hit;   // cached primary hit for sample regeneration (spp > 1)
    float4* di;           // row N4: direct illumination of the primary surface {rgb, valid} (IsDIEnabled only)
    // spp > 1, untextured scenes (null otherwise): what the first shading of a pixel's primary surface computes and every later sample
    // of that pixel would compute again -- {N.xyz, spawn offset} {lobe weights, sphere id} {primary direction, -} per slot.  Written by
    // the primary pass, read by the looping pass when it regenerates a sample (L2-resident: a lane re-reads its 48 bytes every few steps).
    float4* primary_cache;
};

}  // namespace pt
