/* pt_api.h -- C-ABI of the MI355X-native path-tracing hot path (libpt_hip.so).
 *
 * The reference has no plugin/FFI seam; its operator interface for this path is the
 * pass-object pattern.  Each entry point below names the reference interface
 * it replaces (citations into /root/reference):
 *
 *   pt_create / pt_destroy   Raytracing::Raytracing(CommandList&) + GBufferGeneration ctor: PSO / root
 *                            signature / constant-buffer creation   Source/Raytracing.ixx:61-90, GBufferGeneration.ixx:54-68
 *   pt_set_scene             Scene::Load + Refresh -> InstanceData/ObjectData upload
 *                                                       Source/Scene.ixx:123-219, Source/App.cpp:977-1028
 *   pt_build_accel           Scene::CreateAccelerationStructures (BLAS+TLAS via RTXMU)
 *                                                       Source/Scene.ixx:225-284, RaytracingHelpers.ixx:28-74
 *   pt_set_camera            commandList.Copy(*m_GPUBuffers.Camera, {m_camera})   Source/App.cpp:542-553
 *   pt_set_constants         Raytracing::SetConstants(const GraphicsSettings&)    Source/Raytracing.ixx:92-104
 *   pt_render                GBufferGeneration::Render + Raytracing::Render -> Dispatch / DispatchRays(W,H,1)
 *                                                       Source/GBufferGeneration.ixx:80-117, Raytracing.ixx:106-112,228-249
 *   pt_render_tiles / pt_unpack_tiles / pt_set_partition
 *                            (no reference analogue: single adapter) tile partition for multi-GPU, SURVEY 8e
 *   pt_last_error            ThrowIfFailed -> std::system_error text  Source/ErrorHelpers.ixx:16-32
 *
 * Conventions: plain pointers and sizes, status-code errors (no exceptions cross the boundary),
 * opaque context, caller-owned host memory, callee-owned device memory.  A context is not
 * thread-safe (the reference pass objects are driven from the main thread only, App.cpp:565-644).
 * There is NO CPU fallback: pt_create fails with PT_ERR_NO_DEVICE when no HIP device exists.
 */
#ifndef PT_API_H
#define PT_API_H

#include "pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtContext PtContext;

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = 1,
    PT_ERR_NO_DEVICE = 2,
    PT_ERR_HIP = 3,
    PT_ERR_STATE = 4,       /* call order violated (e.g. render before build_accel) */
    PT_ERR_UNSUPPORTED = 5, /* Denoiser requested, a second texture-coordinate set, ... */
    PT_ERR_OOM = 6
} PtStatus;

typedef struct PtConfig {
    int32_t device;         /* HIP device ordinal */
    uint32_t tile_size;     /* multi-GPU tile edge in pixels, a power of two in [8, 1024]; 0 -> 32 */
    uint64_t stream;        /* hipStream_t to run on (e.g. a torch.cuda.Stream's handle); 0 -> context-owned stream, or the
                               legacy default (null) stream with PT_FLAG_DEFAULT_STREAM */
    uint32_t flags;         /* PT_FLAG_* */
    uint32_t frames_in_flight; /* 0 or 1: one frame at a time; 2..8: that many lanes (see PT_FLAG_TWO_FRAMES_IN_FLIGHT).  3 is the
                                  recommended value: up to three lanes run as streams of the highest priority, whose hardware queues
                                  they share with nothing else unless the process creates highest-priority streams of its own; more
                                  lanes share the default-priority queues with every other stream of the process */
} PtConfig;

enum {
    PT_FLAG_NO_LDS_SCENE = 1u,   /* never stage the BVH into LDS (debug / A-B) */
    PT_FLAG_NO_GRAPH = 2u,       /* reserved, no effect: hipGraph replay of the per-frame launches was measured and rejected
                                    (8.4 us host for a 3-kernel graph vs 10.6 us for three launches, slower end to end) */
    PT_FLAG_HOST_LBVH = 4u,      /* build the LBVH on the host instead of on the GPU (debug / A-B) */
    PT_FLAG_SPLIT_KERNELS = 8u,  /* separate traverse / shade kernels with a hit stream instead of the fused bounce kernel */
    PT_FLAG_FAST_BUILD = 64u,    /* always build the device LBVH (the PREFER_FAST_BUILD analogue).  Default: scenes of up to 4096
                                    spheres get a SAH topology built on the host (PREFER_FAST_TRACE, Source/Scene.ixx:247,283) */
    PT_FLAG_DEFAULT_STREAM = 32u, /* with stream == 0: run on the legacy default stream instead of a context-owned one */
    PT_FLAG_TWO_FRAMES_IN_FLIGHT = 16u /* frames in flight (same as PtConfig.frames_in_flight = 2; that field allows up to 8):
                                          consecutive render calls rotate over N internal streams ("lanes", each with its own
                                          work buffers) so that the latency-bound tail of one frame overlaps the start of the
                                          next ones.  The caller must rotate over N output buffers; whatever it queues on
                                          `stream` after a render call is ordered after that frame.  A frame itself is ordered
                                          after what the caller had queued on `stream` before the render call N - 1 calls
                                          earlier (the consumer of its buffer) -- NOT after work queued since then: do not
                                          clear or fill an output buffer on `stream` right before rendering into it (every
                                          pixel and every padding pixel is written by the frame anyway). */
};

typedef struct PtAccelInfo {
    uint32_t leaf_count;     /* == sphere count */
    uint32_t node_count;     /* internal nodes (leaf_count - 1, or 0 for a single sphere) */
    uint32_t depth;          /* max root-to-leaf edge count */
    uint32_t lds_resident;   /* 1 if traversal stages the whole BVH in LDS */
    float bounds_min[3];
    float bounds_max[3];
    float build_ms;          /* device or host build time */
    uint32_t builder;        /* PT_BUILDER_*: which builder made the topology */
} PtAccelInfo;

enum { PT_BUILDER_DEVICE_LBVH = 0, PT_BUILDER_HOST_LBVH = 1, PT_BUILDER_HOST_SAH = 2 };

typedef struct PtStats {
    uint64_t rays;              /* CastRay-equivalents traced, primaries included */
    uint64_t paths;             /* (pixel, sample) pairs started */
    uint64_t pixels;            /* pixels rendered by this call */
    double ms_total;            /* device time of the whole call (HIP events on the context's stream) */
    double ms_traverse;         /* summed device time of the traverse launches (0 unless profiling on) */
    double ms_shade;            /* summed device time of the shade launches (0 unless profiling on) */
    uint32_t traverse_launches;
    uint32_t shade_launches;
    uint64_t bytes_algorithmic; /* DESIGN.md byte model: per-ray queue traffic + per-path accumulate/store */
    double ms_tail;             /* device time of the fused tail launch (0 unless profiling on) */
    uint32_t tail_launches;
    uint32_t beams_used;        /* pt_render*: 1 = the primary pass took its candidates from primary-beam lists (the exact ones of a resting
                                   view, or the widened ones that follow a camera travelling and turning smoothly -- DESIGN.md "Primary beams");
                                   pt_get_totals: the number of such frames since the last reset */
    uint64_t rays_first_pass_inline; /* pt_get_totals only: secondary rays the primary passes traced in registers (the first
                                        bounce of a 1-spp frame never enters a queue); part of `rays` */
    uint64_t node_visits;       /* pt_get_totals only, scenes traversed in global memory: BVH node records read and ... */
    uint64_t sphere_tests;      /* ... sphere records tested by the traversal kernels (the scene term of the byte accounting) */
} PtStats;

/* BVH node as traversed on the device (DESIGN.md "LBVH layout"); exposed for structural tests. */
typedef struct PtBvhNode {
    float lo0[3], hi0[3];   /* child 0 AABB (padded) */
    float lo1[3], hi1[3];   /* child 1 AABB (padded) */
    int32_t child0, child1; /* >= 0: internal node index; < 0: leaf, Morton-sorted sphere index = ~child */
    int32_t parent;         /* -1 for the root */
    int32_t _pad;
} PtBvhNode;

PtStatus pt_create(const PtConfig *config, PtContext **out_ctx);
void pt_destroy(PtContext *ctx);

/* Copies n spheres + n materials (material i belongs to sphere i == ObjectIndex) and the scene constants.  n == 0 is a legal scene
 * (a TLAS without instances; the pointers may be NULL): every ray misses and every pixel is the environment.
 * Materials whose AlphaMode is not Opaque make their sphere non-opaque geometry, as Scene::CreateAccelerationStructures does
 * (Source/Scene.ixx:242-243): every closest-hit query of the path (primary, bounce and shadow rays, pt_trace_rays) then runs
 * TraceRay's candidate loop for it (Shaders/RaytracingHelpers.hlsli:19-43) -- a crossing of the ray with the sphere's surface counts
 * only if IsOpaque accepts it: BaseColor.a, times the alpha of the base-colour map at that crossing once pt_set_textures has
 * given the sphere one, >= AlphaCutoff; the near crossing is tried first, then the far one (DESIGN.md spec S10). */
PtStatus pt_set_scene(PtContext *ctx, const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                      const PtSceneData *scene_data);
/* Builds the LBVH over the current spheres.  info may be NULL. */
PtStatus pt_build_accel(PtContext *ctx, PtAccelInfo *info);
/* Moving spheres (the reference rebuilds its TLAS every frame while the physics runs, Source/App.cpp:605-608):
 * pt_update_spheres uploads new centres / radii for the SAME n objects, pt_refit_accel recomputes every box of the
 * existing tree bottom-up (the topology of the last pt_build_accel is kept; any valid BVH gives identical images, only
 * traversal cost drifts -- call pt_set_scene + pt_build_accel again to rebuild).  Both are asynchronous.  The moved spheres hold for
 * every frame from the next render call on, until they are moved again: the lane of the next frame receives them here, the other lanes
 * take them over (and refit) when they render next.  pt_refit_accel may be omitted -- a render call whose lane has not been refitted since
 * the spheres moved refits by itself; calling it lets the refit start before the render call is made. */
PtStatus pt_update_spheres(PtContext *ctx, const PtSphere *spheres, uint32_t n);
PtStatus pt_refit_accel(PtContext *ctx);
PtStatus pt_set_camera(PtContext *ctx, const PtCamera *camera);
/* Raytracing::SetConstants.  Denoiser must be 0 (Denoiser::None) and IsShaderExecutionReorderingEnabled is ignored.
 * IsDIEnabled = 1 (row N4) adds the sphere-light direct-illumination pass, the build's stand-in for the RTXDI passes
 * whose DI texture Raytracing.hlsl:150-163 reads: one emitter / one cone direction per pixel before the bounce passes,
 * the emission of first-bounce hits reached through a reflective lobe dropped (:302), DI added to the radiance (:381). */
PtStatus pt_set_constants(PtContext *ctx, const PtGraphicsSettings *settings);

/* Render rect (NULL = whole RenderSize) of the frame described by the current constants/camera.
 * out: rect.w*rect.h float4 (r,g,b,1), row-major inside the rect; a HOST pointer if out_is_device == 0
 * (synchronous copy-out), else a DEVICE pointer written on the context's stream (asynchronous).
 * stats may be NULL; requesting stats synchronises the stream. */
PtStatus pt_render(PtContext *ctx, const PtRect *rect, void *out, int out_is_device, PtStats *stats);

/* Multi-GPU tile partition: the RenderSize image is cut into tile_size^2 tiles in row-major
 * tile order; tile t belongs to rank (t % world).  pt_tiles_count returns this rank's tile count for the
 * current RenderSize. */
PtStatus pt_set_partition(PtContext *ctx, uint32_t rank, uint32_t world);
uint32_t pt_tiles_count(PtContext *ctx, uint32_t rank);
/* Render this rank's tiles into a packed DEVICE buffer of pt_tiles_count(rank) * tile_size^2 float4
 * (tile-major, row-major inside a tile; pixels outside the image are zero).  Asynchronous unless stats != NULL. */
PtStatus pt_render_tiles(PtContext *ctx, void *out_device_packed, PtStats *stats);
/* Un-swizzle: given the concatenation [rank 0 tiles | rank 1 tiles | ...] where every rank's block is padded
 * to max_tiles_per_rank tiles (what a gather of equal-sized buffers yields), write the full W*H float4 frame. */
PtStatus pt_unpack_tiles(PtContext *ctx, const void *gathered_device, uint32_t max_tiles_per_rank,
                         void *frame_device);
/* Weighted partition.  xGMI is point-to-point, so the rank that assembles the frame receives every other rank's tiles
 * over one link each while its own tiles cost no transfer: giving it a larger share balances render time against
 * exchange time.  A context owns the tiles t with  first <= t % stride < first + run  (in increasing t; run = 0 owns
 * nothing); pt_set_partition(rank, world) is pt_set_partition_ex(rank, 1, world).  Root weight k over N ranks:
 * stride = N - 1 + k, root (0, k, stride), rank r >= 1 (k - 1 + r, 1, stride).  pt_tiles_count_ex counts a range's tiles
 * for the current RenderSize; pt_render_tiles renders the context's range. */
PtStatus pt_set_partition_ex(PtContext *ctx, uint32_t first, uint32_t run, uint32_t stride);
uint32_t pt_tiles_count_ex(PtContext *ctx, uint32_t first, uint32_t run, uint32_t stride);
/* Un-swizzle n_parts packed buffers laid out part_stride_px float4 apart, part i holding the tiles of the range
 * (first0 + i * run, run, stride).  Pixels of tiles outside these ranges are left untouched, so the frame is assembled
 * by one call for the root's own range and one for the gathered ranges. */
PtStatus pt_unpack_tiles_ex(PtContext *ctx, const void *packed_device, uint64_t part_stride_px, uint32_t n_parts,
                            uint32_t first0, uint32_t run, uint32_t stride, void *frame_device);
/* RGB exchange: every pixel of a frame has alpha 1, so the buffers that cross the links can carry 12 instead of 16 bytes
 * per pixel.  pt_pack_rgb copies n_pixels float4 (packed tiles, any number of frames) to 3 floats per pixel;
 * pt_unpack_tiles_rgb is pt_unpack_tiles_ex for such parts (part_stride_px still counts pixels) and writes alpha = 1. */
PtStatus pt_pack_rgb(PtContext *ctx, const void *src_device, uint64_t n_pixels, void *dst_device);
PtStatus pt_unpack_tiles_rgb(PtContext *ctx, const void *packed_device, uint64_t part_stride_px, uint32_t n_parts,
                             uint32_t first0, uint32_t run, uint32_t stride, void *frame_device);

/* Row N1 -- textured spheres: EvaluateMaterial's texture branches + normal mapping (Shaders/ShadingHelpers.hlsli:53-103,
 * 161-235) over analytic sphere UVs / tangents (csrc/pt_texture.h).  Replaces the texture part of Scene::Load
 * (Source/Scene.ixx:123-180: per-object Textures -> TextureMapInfoArray in ObjectData).  Call after pt_set_scene:
 *   textures[n_textures]   decoded images (copied; converted to linear float4 on upload)
 *   object_textures[n]     one TextureMapInfoArray per sphere (n = the scene's sphere count); Descriptor < n_textures or ~0u;
 *                          NULL = no sphere has maps (the table then only holds the environment map)
 *   rotations              n unit quaternions (x, y, z, w): object -> world rotation of each sphere, NULL = identity
 * n_textures == 0 removes all textures.  Every kernel that shades has a textured variant, selected per launch.
 * The table is also where SceneData.EnvironmentLightTextureDescriptor points (row a18's texture branch,
 * ShadingHelpers.hlsli:13-24: a lat-long map, or the first of the six faces of a cube map; usually PT_TEXTURE_RGBA32_FLOAT);
 * pt_render* fails with PT_ERR_STATE while the scene names an environment texture the table does not hold.
 * pt_update_rotations replaces the quaternions (Earth's spin, the Moon's tidal lock: Source/MyScene.ixx:240-291).  It does no
 * device work and does not wait: the frames in flight keep the rotations they were submitted with, every later render call
 * uploads the new ones into its lane's own copy on its own stream.  A base-colour map on a sphere whose AlphaMode is not
 * Opaque makes its alpha test a per-crossing one (pt_set_scene). */
PtStatus pt_set_textures(PtContext *ctx, const PtTexture *textures, uint32_t n_textures,
                         const PtObjectTextures *object_textures, const float *rotations);
PtStatus pt_update_rotations(PtContext *ctx, const float *rotations, uint32_t n);

/* Row N3 -- display transform and progressive accumulation, on the context's stream (asynchronous).
 * pt_tonemap replaces App::Impl::ToneMap (Source/App.cpp:1731-1757, DirectXTK ToneMapPostProcess): hdr = n_pixels float4
 * (r,g,b,_) -> out = n_pixels packed uint32: R8G8B8A8_UNORM for the Linear / SRGB transfer functions, R10G10B10A2_UNORM
 * for ST2084.  Both pointers are DEVICE pointers.
 * pt_accumulate keeps the running mean of successive frames (progressive refinement while the camera rests; the
 * reference accumulates inside its denoisers, which are out of scope): accum = frames_accumulated == 0 ? radiance
 * : accum + (radiance - accum) / (frames_accumulated + 1), per channel, alpha included. */
PtStatus pt_tonemap(PtContext *ctx, const void *hdr_device, uint32_t n_pixels, const PtToneMapParams *params,
                    void *out_device);
PtStatus pt_accumulate(PtContext *ctx, void *accum_device, const void *radiance_device, uint32_t n_pixels,
                       uint32_t frames_accumulated);

/* Test / tooling hooks. */
/* Closest hit of n rays against the scene and accel of the last pt_set_scene / pt_build_accel (spheres moved by pt_update_spheres live in
 * the lanes' private copies and are not seen here): o,d = n*3 floats (d unit length), tmin per call.
 * Outputs host arrays t[n], id[n] (id = 0xFFFFFFFF on miss).  use_bvh = 0 runs the device brute-force kernel. */
PtStatus pt_trace_rays(PtContext *ctx, const float *origins, const float *directions, uint32_t n, float tmin,
                       int use_bvh, float *out_t, uint32_t *out_id);
/* As pt_trace_rays through the LBVH, additionally returning per ray {internal nodes visited, spheres tested}
 * (out_visits: n * 2 uint32). */
PtStatus pt_trace_rays_stats(PtContext *ctx, const float *origins, const float *directions, uint32_t n, float tmin,
                             float *out_t, uint32_t *out_id, uint32_t *out_visits);
/* Copy the device BVH to host: nodes[node_count]; leaf child c < 0 refers to Morton-sorted index ~c, whose
 * original sphere id is sorted_id[~c] (pt_accel_download_order: sorted_id[leaf_count]). */
PtStatus pt_accel_download(PtContext *ctx, PtBvhNode *nodes, uint32_t capacity);
PtStatus pt_accel_download_order(PtContext *ctx, uint32_t *sorted_id, uint32_t capacity);
/* Host LBVH builder (the PT_FLAG_HOST_LBVH path), callable without a context or a GPU, for structural tests:
 * nodes[n-1], sorted_id[n]; returns the tree depth through *depth. */
PtStatus pt_lbvh_build_host(const PtSphere *spheres, uint32_t n, PtBvhNode *nodes, uint32_t *sorted_id, uint32_t *depth);
/* Host SAH builder (the topology pt_build_accel gives small scenes), same outputs. */
PtStatus pt_sah_build_host(const PtSphere *spheres, uint32_t n, PtBvhNode *nodes, uint32_t *sorted_id, uint32_t *depth);
/* Turn per-launch hipEvent profiling on/off (an event pair around every kernel launch, on the stream it runs on). */
PtStatus pt_set_profiling(PtContext *ctx, int enabled);
/* Sum of the per-launch event times recorded since profiling was switched on / last reset, over every render call
 * (synchronises): ms_traverse / traverse_launches = compacting passes (and split-schedule primary / traverse launches),
 * ms_shade / shade_launches = split-schedule shade launches, ms_tail / tail_launches = looping passes. */
PtStatus pt_get_profile(PtContext *ctx, PtStats *profile, int reset);
/* Running totals over every render call since the last reset, accumulated on the device without host
 * synchronisation (rays, paths, pixels, bytes_algorithmic; the timing fields are zero).  Synchronises the stream. */
PtStatus pt_get_totals(PtContext *ctx, PtStats *totals, int reset);
/* Queue sizes of the last spp == 1 frame: sizes[k] = rays in queue k (sizes[0] = path slots).  Returns the number of
 * valid entries through *n (0 if none).  Synchronises the stream. */
PtStatus pt_get_queue_sizes(PtContext *ctx, uint32_t *sizes, uint32_t capacity, uint32_t *n);
/* Device buffers for callers that do not link a GPU runtime themselves (a C++ host written against this header only): the
 * packed tile buffers of pt_render_tiles / pt_gather and the assembled frames of pt_unpack_tiles live in such memory.  The
 * reference's counterpart is the app-owned GPUBuffer / Texture objects handed to the passes (Source/App.cpp:366-368).
 * pt_device_free waits for the frames in flight; pt_download copies device -> host on the context's stream and waits. */
PtStatus pt_device_alloc(PtContext *ctx, uint64_t bytes, void **out_device);
PtStatus pt_device_free(PtContext *ctx, void *device);
PtStatus pt_download(PtContext *ctx, const void *device, void *host, uint64_t bytes);

/* Multi-GPU exchange of HDR tiles (SURVEY 8b `pt_gather`, 8e): one process per GPU, each with its own context; the frame is
 * tile-partitioned (pt_set_partition[_ex] / pt_render_tiles) and the packed tile buffers are gathered to the rank that assembles
 * it.  The reference renders on ONE adapter (Source/DeviceResources.cpp:507 picks a single DXGI adapter); this is the path's
 * multi-GPU extension.  Collectives are RCCL over xGMI; librccl.so is loaded at run time by pt_comm_unique_id / pt_comm_init
 * (PT_ERR_UNSUPPORTED when it cannot be), so single-GPU hosts do not depend on it.
 *   pt_comm_unique_id  rank 0 creates the 128-byte id and ships it to the other ranks (file, socket, launcher environment).
 *   pt_comm_init       collective over all `world` ranks (ncclCommInitRank); the context owns the communicator.
 *   pt_gather          on the context's stream, ordered after the render calls queued before it: every rank other than `root`
 *                      sends `bytes` bytes from send_device (recv_device ignored); the root receives world - 1 parts, the part of
 *                      rank r at recv_device + (r < root ? r : r - 1) * bytes (send_device ignored: its own tiles never travel).
 *                      One grouped ncclSend/ncclRecv exchange, so all inbound links of the root are used at once. */
#define PT_COMM_ID_BYTES 128
PtStatus pt_comm_unique_id(void *id_out);
PtStatus pt_comm_init(PtContext *ctx, const void *id, uint32_t rank, uint32_t world);
PtStatus pt_comm_destroy(PtContext *ctx);
PtStatus pt_gather(PtContext *ctx, const void *send_device, void *recv_device, uint64_t bytes, uint32_t root);

PtStatus pt_synchronize(PtContext *ctx);

const char *pt_last_error(PtContext *ctx);
const char *pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
