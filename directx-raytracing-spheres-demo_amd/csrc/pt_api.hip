// pt_api.hip -- implementation of the C-ABI (include/pt_api.h) on the HIP runtime: context, device
// memory, LBVH upload, and the per-frame launch sequence of the wavefront kernel set.
// There is no CPU fallback here by design: without a HIP device pt_create fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is loaded at run time (pt_comm_init)
#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <limits>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_kernels.h"
#include "pt_lbvh.h"
#include "pt_lbvh_gpu.h"

using namespace pt;

namespace {

constexpr uint32_t kMaxLdsBytes = 160u * 1024u;  // gfx950 LDS per CU / per workgroup
constexpr uint32_t kSahMaxSpheres = 4096u;        // host SAH topology up to here, device LBVH above (pt_build_accel)
constexpr uint32_t kLdsSceneBudget = 64u * 1024u; // stage the BVH in LDS only while two 512-thread blocks still fit per CU

struct EventPair {
    hipEvent_t a, b;
    int kind;  // 0 primary, 1 traverse / fused bounce, 2 shade, 3 looping pass
};

}  // namespace

// PT_* tuning knobs (DESIGN.md "Tuning knobs"): environment variables for A/B runs, read ONCE when the context is created --
// the render path never touches the environment.  -1 = not set (the measured default applies).
struct Knobs {
    int split = -1, traverse_blocks_per_cu = -1, fused_threads = -1, no_adaptive_grid = -1, shade_blocks_per_cu = -1, tail_threshold = -1,
        tail_blocks_per_cu = -1, loop_threads = -1, inline2_min_slots = -1, tail_after = -1, seg = -1, loop_use_tail = -1, fuse_loop = -1,
        ray_replacement = -1, dyn_blocks_per_cu = -1, debug_counts = -1, sah = -1, sah_max_spheres = -1, beams = -1, wide = -1, descent = -1, roctx = -1, lane_priority = -1, fused_refit = -1, beam_reach = -1, beam_max_slack_pct = -1, beam_max_margin = -1, beam_share_wgs = -1;
};

// Per-frame-in-flight state (see PtContext::lanes).
struct Lane {
    hipStream_t stream = nullptr;  // == PtContext::stream when there is a single lane
    hipEvent_t ev_done = nullptr;
    size_t cap_slots = 0;
    RayQueue q[2]{};
    Scratch scratch{};
    bool scratch_spp = false;
    uint32_t* d_counts = nullptr;  // two parities: [0, cap_counts) and [cap_counts, 2 cap_counts)
    size_t cap_counts = 0;
    uint32_t parity = 0;           // parity of the frame being (or last) submitted on this lane
    uint32_t* h_counts = nullptr;  // pinned
    // queue sizes of a recent frame (pinned, written by an async copy, read without waiting): they only size the
    // launch grids -- every kernel is a grid-stride loop, so a stale or missing estimate costs time, never correctness
    uint32_t* h_prev_counts = nullptr;   // host-mapped: the GPU writes it when it folds a frame's counters (no copy call)
    uint32_t* d_prev_counts = nullptr;   // device address of h_prev_counts
    uint64_t prev_signature = 0;
    uint32_t* d_seg_counts = nullptr;        // kMaxSegs segment sizes of the primary pass -> looping pass hand-over
    unsigned long long* d_totals = nullptr;  // [0] running secondary-ray total, [1] last folded frame, [2],[3] tail counters,
                                             // [4] running count of in-register secondary rays of primary passes, [5] unused,
                                             // [6] node visits, [7] sphere tests of the global-memory traversal kernels
    // private copy of the moving part of the scene (pt_update_spheres / pt_refit_accel): spheres, Morton-ordered spheres
    // and node boxes; null = this lane renders the context's master scene
    float4* d_sph = nullptr;
    float4* d_sph_sorted = nullptr;
    float4* d_nodes = nullptr;
    uint32_t* d_refit_flags = nullptr;
    uint32_t* d_refit_hdr = nullptr;
    PtSphere* h_stage = nullptr;      // pinned upload staging, host-mapped ...
    const float4* d_stage = nullptr;  // ... and its device address (the single-launch refit of small scenes reads the staging buffer itself)
    hipEvent_t ev_upload = nullptr;   // the last upload from h_stage has been consumed
    hipEvent_t ev_poll[4] = {};       // queue-size read-backs of the last passes (spp > 1 lagged polling)
    uint32_t scene_n = 0;             // sphere count the private copy was allocated for
    bool scene_private = false;
    uint64_t sph_gen = 0;         // the pt_update_spheres generation this lane's private scene holds (PtContext::sph_gen)
    bool needs_refit = false;     // spheres were staged on this lane and its boxes / Morton-ordered copy have not been redone yet
    bool upload_pending = false;  // h_stage holds spheres that have not been copied to d_sph yet (pt_update_spheres of a small scene: pt_refit_accel's kernel reads them)
    const void* last_out = nullptr;   // output buffer of the lane's latest frame (render_common: repeated buffers inside the window)
    // object rotations (textured scenes): the lane's own copy, refreshed from PtContext::h_rot when its generation is behind
    float4* d_rot = nullptr;
    float4* h_rot_stage = nullptr;    // pinned
    hipEvent_t ev_rot = nullptr;      // the last upload from h_rot_stage has been consumed
    uint64_t rot_gen = 0;
    uint32_t rot_n = 0;
};
constexpr uint32_t kMaxLanes = 8;

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t flags = 0;
    uint32_t tile_size = 32;
    uint32_t num_cus = 256;
    Knobs knobs;
    std::string err;

    // scene
    uint32_t n = 0;
    float4* d_sph = nullptr;
    float4* d_mats = nullptr;
    PtSceneData sd{};
    std::vector<PtSphere> h_sph;
    bool scene_set = false;

    // textures (row N1): table of linear float4 images + per-sphere map indices and rotations
    std::vector<float4*> d_tex_images;
    std::vector<std::pair<uint32_t, uint32_t>> tex_dims;  // width, height of every table entry
    TexView* d_tex = nullptr;
    uint32_t* d_tex_maps = nullptr;  // n * 8
    float4* d_rot = nullptr;         // n: the rotations as of pt_set_textures (single-lane contexts update it in stream order)
    std::vector<float4> h_rot;       // latest rotations (pt_update_rotations); lanes pick them up when they next render
    uint64_t rot_gen = 0;            // generation of h_rot (never reset) ...
    uint64_t rot_master_gen = 0;     // ... and the generation d_rot holds: while they are equal every lane reads d_rot
    bool has_textures = false;

    // emissive spheres (row N4)
    uint32_t* d_lights = nullptr;
    uint32_t n_lights = 0;

    // alpha-tested hits (spec S10): the spheres whose AlphaMode is not Opaque, their class per sphere on the device (null while
    // every sphere is kAlphaVisible) and the leaf ids carrying it (Morton order; null = the traversal reads d_sorted_id)
    struct AlphaMat { uint32_t id; float base[4]; float cutoff; uint32_t base_map; };
    std::vector<AlphaMat> alpha_mats;
    uint32_t* d_alpha_class = nullptr;
    uint32_t* d_leaf_ids = nullptr;
    bool alpha_tested = false;

    // accel
    float4* d_nodes = nullptr;
    float4* d_wide = nullptr;        // 4-wide view of the tree (global-memory scenes only; null otherwise)
    float4* d_sph_sorted = nullptr;
    uint32_t* d_sorted_id = nullptr;
    uint32_t n_nodes = 0, depth = 0;
    bool lds_scene = false;
    bool accel_valid = false;
    LbvhResult lbvh;  // host copy (download / info); filled by either builder
    LbvhGpu* gpu_builder = nullptr;

    // frame state
    PtCamera cam{};
    PtGraphicsSettings gs{};
    bool cam_set = false, gs_set = false;
    uint32_t rank = 0, world = 1;                       // pt_set_partition (kept for pt_tiles_count(rank))
    uint32_t part_first = 0, part_run = 1, part_stride = 1;  // the residue range this context renders (pt_set_partition_ex)

    // work buffers: one set per frame in flight.  Frame f runs on lane f % n_lanes, on that lane's own stream, so the
    // latency-bound looping pass of one frame overlaps the throughput-bound first passes of the next.
    Lane lanes[kMaxLanes];
    uint32_t n_lanes = 1;
    uint32_t next_lane = 0;
    uint32_t last_lane = 0;
    hipEvent_t ev_in[kMaxLanes] = {};  // markers on `stream` at the start of the last n_lanes render calls
    uint64_t calls = 0;
    uint64_t sph_gen = 0;      // counts pt_update_spheres calls since pt_set_scene; latest_lane: the lane whose staging buffer holds the newest spheres
    int latest_lane = -1;
    bool empty_scene = false;  // pt_set_scene(n = 0): one internal sphere that no ray can hit stands in (see pt_set_scene)
    float4* d_out = nullptr;
    size_t cap_out = 0;
    uint64_t tot_pixels = 0, tot_paths = 0, tot_fixed_bytes = 0, tot_sec_coeff = 96;  // host-known parts of the totals
    uint32_t tot_beam_frames = 0;  // frames since the last reset whose primary pass used the primary-beam lists

    // Primary beams (DESIGN.md "Primary beams"): per-8x8-block candidate sphere lists for the primary pass.  They depend on the
    // camera's lens, the frame geometry and the scene, on the camera's POSITION up to the slack (Beam::slack) and on its ORIENTATION
    // up to the pixel margin (make_beam) they were built with -- not on the frame index or the jitter (the beams are a pixel wider
    // than the blocks).  A view that RESTS gets exact lists on its second frame (one launch on a side stream); a camera that
    // travels and turns gets lists centred and oriented some frames ahead of it, with a slack of a few frames' travel and a margin
    // of a few frames' turn, built in shares inside the frames' own primary passes
    // while the frames use the previous ones -- a frame never waits for a build of the moving kind: it takes the newest lists that
    // are readable and hold for its pose, or traverses per ray.
    struct BeamLists {
        uint32_t* d_lists = nullptr;   // n_blocks records of 16 dwords
        size_t cap_blocks = 0;
        std::vector<uint32_t> key;     // orientation, frame geometry, scene generation of the lists in d_lists; empty = none
        float pos[3] = { 0, 0, 0 };    // the camera position they were built around ...
        float slack = 0.0f;            // ... and how far from it they hold
        float basis[9] = {};           // the orientation (Right, Up, Forward) they were built for ...
        float margin_px = 0.0f;        // ... and by how many pixels a ray's crossing of the image may differ from that orientation's
        hipEvent_t ev_ready = nullptr;   // the build has finished
        bool building = false;         // launched, ev_ready not yet seen complete
        bool used = false;             // read by a frame since the build (a rebuild must wait for the lanes)
        uint64_t first_call = 0;       // the first render call whose frame may read them (BeamCache::calls)
        uint64_t last_use_call = 0;    // the last render call whose frame was handed them
        uint64_t built_call = 0;       // the render call that started (resting view) or completed (moving camera) their build
    };
    struct BeamCache {
        BeamLists buf[2];
        int cur = 0;                     // the lists frames use; the other buffer is the one a build goes to
        uint64_t calls = 0;              // render calls that consulted the cache
        std::vector<uint32_t> last_key;  // key / position of the previous render call
        float last_pos[3] = { 0, 0, 0 };
        hipStream_t stream = nullptr;      // side stream (the builds of resting views)
        hipEvent_t ev_last_use = nullptr;  // scratch event of a rebuild (orders it after the lanes' frames in flight)
        // a moving camera's next lists, built a share per frame inside the frames' primary passes (FrameParams::beam_job)
        struct { bool active = false; BeamLists* dst = nullptr; std::vector<uint32_t> key; float centre[3] = { 0, 0, 0 }; float slack = 0.0f; float basis[9] = {}; float margin_px = 0.0f; uint32_t next_block = 0, n_blocks = 0; } inc;
        float last_vel[3] = { 0, 0, 0 };   // the camera's travel between the two calls before this one (its change bounds how far to trust the extrapolation)
        double last_turn[3] = { 0, 0, 0 }; // ... and its turn (rotation vector)
        float last_basis[9] = {};          // orientation of the previous render call
        bool have_vel = false;
    } beam;
    uint64_t scene_gen = 0;  // bumped by everything that changes what a ray can hit
    float min_radius = 0.0f;  // smallest sphere of the scene set by pt_set_scene (bounds the slack of a moving camera's beam lists)

    // multi-GPU exchange (pt_comm_init / pt_gather): the RCCL communicator of this rank
    ncclComm_t comm = nullptr;
    uint32_t comm_rank = 0, comm_world = 1;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profiling = false;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
};

namespace {

// roctx ranges (the reference brackets its passes with PIX events): with PT_ROCTX=1 every render call and BVH build /
// refit is a named range on rocprofv3's marker timeline (rocprofv3 --marker-trace --kernel-trace).  libroctx64 is resolved at run
// time, like RCCL; without the knob nothing is loaded or called.
struct Roctx {
    void* handle = nullptr;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};

Roctx& roctx()
{
    static Roctx r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char* name : { "librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1" }) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (r.handle) {
        r.push = reinterpret_cast<int (*)(const char*)>(dlsym(r.handle, "roctxRangePushA"));
        r.pop = reinterpret_cast<int (*)()>(dlsym(r.handle, "roctxRangePop"));
        if (!r.push || !r.pop) { r.push = nullptr; r.pop = nullptr; }
    }
    return r;
}

struct RoctxRange {
    bool on = false;
    RoctxRange(const PtContext* c, const char* name)
    {
        if (c && c->knobs.roctx > 0 && roctx().push) { (void)roctx().push(name); on = true; }
    }
    ~RoctxRange() { if (on) (void)roctx().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// RCCL is resolved at run time, on first use: a single-GPU host never needs it, and inside a process that already carries an
// RCCL (PyTorch's) the loader hands back that very copy (same SONAME), so one collective library serves the process.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    // A process that already carries an RCCL (PyTorch bundles its own copy, under its own path) must not get a second one: look through
    // the objects that are mapped for a librccl and take a handle to THAT one (RTLD_NOLOAD); only a process without any loads the system's.
    std::string mapped;
    dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) {
        if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) { *static_cast<std::string*>(data) = info->dlpi_name; return 1; }
        return 0;
    }, &mapped);
    if (!mapped.empty()) r.handle = dlopen(mapped.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        if (r.handle) break;
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) { r.error = std::string("cannot load librccl.so: ") + dlerror(); return r; }
    auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.error.empty()) { dlclose(r.handle); r.handle = nullptr; }
    return r;
}

PtStatus fail(PtContext* ctx, PtStatus st, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return st;
}

#define PT_HIP(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP,                      \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                            \
        }                                                                                              \
    } while (0)

template <typename T>
void free_dev(T*& p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

void free_lane_scene(Lane& L)
{
    free_dev(L.d_sph); free_dev(L.d_sph_sorted); free_dev(L.d_nodes); free_dev(L.d_refit_flags); free_dev(L.d_refit_hdr);
    if (L.h_stage) { (void)hipHostFree(L.h_stage); L.h_stage = nullptr; L.d_stage = nullptr; }
    L.scene_n = 0;
    L.scene_private = false;
    L.upload_pending = false;
    L.needs_refit = false;
}

void free_lane_buffers(Lane& L)
{
    for (auto& q : L.q) { free_dev(q.q0); free_dev(q.q1); free_dev(q.q2); free_dev(q.hit); }
    free_dev(L.scratch.sample_rad); free_dev(L.scratch.radiance); free_dev(L.scratch.primary_hit); free_dev(L.scratch.di); free_dev(L.scratch.primary_cache);
    L.cap_slots = 0;
    L.scratch_spp = false;
}

PtStatus ensure_buffers(PtContext* c, Lane& L, size_t n_slots, bool need_spp, bool need_hits, size_t n_counts, bool need_di = false)
{
    if (n_slots > L.cap_slots || (need_hits && !L.q[0].hit)) {
        PT_HIP(c, hipStreamSynchronize(L.stream));
        free_lane_buffers(L);
        for (auto& q : L.q) {
            PT_HIP(c, hipMalloc(&q.q0, n_slots * sizeof(float4)));
            PT_HIP(c, hipMalloc(&q.q1, n_slots * sizeof(float4)));
            PT_HIP(c, hipMalloc(&q.q2, n_slots * sizeof(float4)));
            if (need_hits) PT_HIP(c, hipMalloc(&q.hit, n_slots * sizeof(uint2)));  // split schedule only
        }
        PT_HIP(c, hipMalloc(&L.scratch.sample_rad, n_slots * sizeof(float4)));
        L.cap_slots = n_slots;
    }
    if (need_di && !L.scratch.di) PT_HIP(c, hipMalloc(&L.scratch.di, L.cap_slots * sizeof(float4)));
    if ((need_spp || need_di) && !L.scratch.primary_hit) PT_HIP(c, hipMalloc(&L.scratch.primary_hit, L.cap_slots * sizeof(uint2)));
    if (need_spp && !L.scratch_spp) {
        PT_HIP(c, hipMalloc(&L.scratch.radiance, L.cap_slots * sizeof(float4)));
        PT_HIP(c, hipMalloc(&L.scratch.primary_cache, L.cap_slots * 3u * sizeof(float4)));  // Scratch::primary_cache: 48 B per slot
        L.scratch_spp = true;
    }
    if (n_counts > L.cap_counts) {
        free_dev(L.d_counts);
        if (L.h_counts) { (void)hipHostFree(L.h_counts); L.h_counts = nullptr; }
        if (L.h_prev_counts) { (void)hipHostFree(L.h_prev_counts); L.h_prev_counts = nullptr; }
        PT_HIP(c, hipHostMalloc(&L.h_prev_counts, n_counts * sizeof(uint32_t), hipHostMallocMapped));
        std::memset(L.h_prev_counts, 0, n_counts * sizeof(uint32_t));
        PT_HIP(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&L.d_prev_counts), L.h_prev_counts, 0));
        L.prev_signature = 0;
        PT_HIP(c, hipMalloc(&L.d_counts, 4 * n_counts * sizeof(uint32_t)));  // 2 parities x (queue sizes + work cursors)
        PT_HIP(c, hipMemsetAsync(L.d_counts, 0, 4 * n_counts * sizeof(uint32_t), L.stream));
        PT_HIP(c, hipHostMalloc(&L.h_counts, n_counts * sizeof(uint32_t)));
        L.cap_counts = n_counts;
    }
    return PT_OK;
}

// wait for every frame in flight
hipError_t sync_all(PtContext* c)
{
    for (uint32_t i = 0; i < c->n_lanes; i++)
        if (c->lanes[i].stream) { hipError_t e = hipStreamSynchronize(c->lanes[i].stream); if (e != hipSuccess) return e; }
    if (c->beam.stream) { hipError_t e = hipStreamSynchronize(c->beam.stream); if (e != hipSuccess) return e; }
    return c->stream ? hipStreamSynchronize(c->stream) : hipSuccess;
}

void free_textures(PtContext* c)
{
    for (auto& img : c->d_tex_images) free_dev(img);
    c->d_tex_images.clear();
    c->tex_dims.clear();
    free_dev(c->d_tex); free_dev(c->d_tex_maps); free_dev(c->d_rot);
    c->has_textures = false;
}

// Object rotations follow the frames asynchronously: pt_update_rotations only replaces the host copy and bumps a generation;
// the lane that renders next uploads them into ITS copy on ITS stream (pinned staging, no wait for the other frames in flight).
PtStatus sync_lane_rotations(PtContext* c, Lane& L)
{
    if (!c->has_textures || c->rot_gen == c->rot_master_gen || (L.rot_gen == c->rot_gen && L.rot_n == c->n)) return PT_OK;
    const uint32_t n = c->n;
    if (L.rot_n != n) {
        PT_HIP(c, hipStreamSynchronize(L.stream));
        free_dev(L.d_rot);
        if (L.h_rot_stage) { (void)hipHostFree(L.h_rot_stage); L.h_rot_stage = nullptr; }
        PT_HIP(c, hipMalloc(&L.d_rot, (size_t)n * sizeof(float4)));
        PT_HIP(c, hipHostMalloc(&L.h_rot_stage, (size_t)n * sizeof(float4)));
        if (!L.ev_rot) PT_HIP(c, hipEventCreateWithFlags(&L.ev_rot, hipEventDisableTiming));
        L.rot_n = n;
    } else {
        PT_HIP(c, hipEventSynchronize(L.ev_rot));  // the previous upload from the staging buffer has been consumed
    }
    std::memcpy(L.h_rot_stage, c->h_rot.data(), (size_t)n * sizeof(float4));
    PT_HIP(c, hipMemcpyAsync(L.d_rot, L.h_rot_stage, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, L.stream));
    PT_HIP(c, hipEventRecord(L.ev_rot, L.stream));
    L.rot_gen = c->rot_gen;
    return PT_OK;
}

// Alpha-tested hits (pt_device.h): classify the non-opaque spheres from their materials and base-colour maps, and (when the tree
// exists) rebuild the flagged leaf ids.  Called, with nothing in flight, whenever materials, maps or the tree change.
PtStatus refresh_leaf_ids(PtContext* c)
{
    free_dev(c->d_leaf_ids);
    if (!c->d_alpha_class || !c->accel_valid) return PT_OK;
    PT_HIP(c, hipMalloc(&c->d_leaf_ids, (size_t)c->n * sizeof(uint32_t)));
    PT_HIP(c, launch_leaf_ids(c->d_sorted_id, c->d_alpha_class, c->n, c->d_leaf_ids, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

PtStatus update_alpha_classes(PtContext* c)
{
    std::vector<uint32_t> cls;
    bool any = false, tested = false;
    for (const auto& am : c->alpha_mats) {
        // EvaluateBaseColor samples the map when any component of the float4 BaseColor is positive (ShadingHelpers.hlsli:61-72)
        const bool sampled = c->has_textures && am.base_map != ~0u && (am.base[0] > 0.0f || am.base[1] > 0.0f || am.base[2] > 0.0f || am.base[3] > 0.0f);
        const uint32_t k = sampled ? kAlphaTested : (am.base[3] >= am.cutoff ? kAlphaVisible : kAlphaInvisible);
        if (k != kAlphaVisible) {
            if (cls.empty()) cls.assign(c->n, kAlphaVisible);
            cls[am.id] = k;
            any = true;
            tested = tested || k == kAlphaTested;
        }
    }
    free_dev(c->d_alpha_class);
    c->alpha_tested = tested;
    if (any) {
        PT_HIP(c, hipMalloc(&c->d_alpha_class, (size_t)c->n * sizeof(uint32_t)));
        PT_HIP(c, hipMemcpy(c->d_alpha_class, cls.data(), (size_t)c->n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->scene_gen++;  // what a ray can hit has changed (primary-beam lists)
    return refresh_leaf_ids(c);
}

PtStatus validate_frame(PtContext* c)
{
    if (!c->scene_set) return fail(c, PT_ERR_STATE, "pt_set_scene has not been called");
    if (!c->accel_valid) return fail(c, PT_ERR_STATE, "pt_build_accel has not been called for the current scene");
    if (!c->cam_set) return fail(c, PT_ERR_STATE, "pt_set_camera has not been called");
    if (!c->gs_set) return fail(c, PT_ERR_STATE, "pt_set_constants has not been called");
    return PT_OK;
}

inline uint32_t knob_or(int v, uint32_t dflt) { return v < 0 ? dflt : (uint32_t)v; }

// per-lane traversal-stack entries: one per level for the binary walk; the wide walk pushes up to three per wide level
uint32_t stack_entries(const PtContext* c, bool wide)
{
    return wide ? std::max(1u, c->depth) + (c->depth + 1u) / 2u + 2u : std::max(1u, c->depth);
}

SceneView make_scene_view(const PtContext* c, const Lane* L = nullptr)
{
    SceneView sv{};
    const bool priv = L && L->scene_private;
    sv.nodes = priv ? L->d_nodes : c->d_nodes;
    sv.wide = priv ? nullptr : c->d_wide;  // (a lane's refitted private tree is walked through its binary records)
    sv.sph_sorted = priv ? L->d_sph_sorted : c->d_sph_sorted;
    sv.sorted_id = c->d_leaf_ids ? c->d_leaf_ids : c->d_sorted_id;
    sv.alpha_class = c->d_alpha_class;
    sv.alpha_tested = c->alpha_tested ? 1u : 0u;
    sv.sph = priv ? L->d_sph : c->d_sph;
    sv.mats = c->d_mats;
    sv.n = c->n;
    sv.n_nodes = c->n_nodes;
    sv.stack_depth = stack_entries(c, sv.wide != nullptr);
    // Global-memory scenes: a lane makes at most this many node visits before its wave turns to the sphere tests.  Unbounded, the
    // lanes that already hold a leaf idle until the longest descent of the wave ends (hundreds of visits in the 2^20-sphere scene's
    // heavy tail): 3.99 -> 2.54 ms per frame there with any bound from 2 to 24; LDS-resident scenes (coherent, shallow) gain nothing.
    sv.descent_cap = c->lds_scene ? 0u : knob_or(c->knobs.descent, 8u);
    sv.lds_scene = c->lds_scene ? 1u : 0u;
    for (int i = 0; i < 4; i++) sv.env[i] = c->sd.EnvironmentLightColor[i];
    if (c->has_textures) { sv.tex = c->d_tex; sv.tex_maps = c->d_tex_maps; sv.rot = (c->rot_gen != c->rot_master_gen && L && L->d_rot && L->rot_gen == c->rot_gen) ? L->d_rot : c->d_rot; }
    sv.env_tex = c->sd.EnvironmentLightTextureDescriptor;  // ~0u == kNoTexture; render_common has checked it against the table
    sv.env_cube = c->sd.IsEnvironmentLightTextureCubeMap ? 1u : 0u;
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) sv.env_xf[3 * r + k] = c->sd.EnvironmentLightTransform[4 * r + k];
    sv.lights = c->d_lights; sv.n_lights = c->n_lights;
    return sv;
}

FrameParams make_frame_params(const PtContext* c)
{
    FrameParams fp{};
    fp.cam = camera_params(c->cam, c->gs.RenderSize[0], c->gs.RenderSize[1]);
    fp.frame_index = c->gs.FrameIndex;
    fp.bounces = c->gs.Bounces;
    fp.spp = c->gs.SamplesPerPixel;
    fp.rr_enabled = c->gs.IsRussianRouletteEnabled ? 1u : 0u;
    fp.throughput_threshold = c->gs.ThroughputThreshold;
    fp.inv_spp = 1.0f / (float)c->gs.SamplesPerPixel;
    fp.di_enabled = (c->gs.IsDIEnabled && c->n_lights > 0) ? 1u : 0u;  // no emitters: DI is 0 everywhere, nothing to do
    return fp;
}

int env_knob(const char* name)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return -1;
    return (int)std::strtoul(v, nullptr, 10);
}

Knobs read_knobs()
{
    Knobs k;
    k.split = env_knob("PT_SPLIT"); k.traverse_blocks_per_cu = env_knob("PT_TRAVERSE_BLOCKS_PER_CU"); k.fused_threads = env_knob("PT_FUSED_THREADS");
    k.no_adaptive_grid = std::getenv("PT_NO_ADAPTIVE_GRID") ? 1 : -1; k.shade_blocks_per_cu = env_knob("PT_SHADE_BLOCKS_PER_CU");
    k.tail_threshold = env_knob("PT_TAIL_THRESHOLD"); k.tail_blocks_per_cu = env_knob("PT_TAIL_BLOCKS_PER_CU"); k.loop_threads = env_knob("PT_LOOP_THREADS");
    k.inline2_min_slots = env_knob("PT_INLINE2_MIN_SLOTS"); k.tail_after = env_knob("PT_TAIL_AFTER"); k.seg = env_knob("PT_SEG");
    k.loop_use_tail = std::getenv("PT_LOOP_USE_TAIL") ? 1 : -1; k.fuse_loop = env_knob("PT_FUSE_LOOP"); k.beam_reach = env_knob("PT_BEAM_REACH"); k.beam_max_slack_pct = env_knob("PT_BEAM_MAX_SLACK_PCT"); k.beam_max_margin = env_knob("PT_BEAM_MAX_MARGIN"); k.beam_share_wgs = env_knob("PT_BEAM_SHARE_WGS"); k.lane_priority = env_knob("PT_LANE_PRIORITY"); k.fused_refit = env_knob("PT_FUSED_REFIT"); k.ray_replacement = env_knob("PT_RAY_REPLACEMENT");
    k.dyn_blocks_per_cu = env_knob("PT_DYN_BLOCKS_PER_CU"); k.debug_counts = std::getenv("PT_DEBUG_COUNTS") ? 1 : -1; k.sah = env_knob("PT_SAH");
    k.sah_max_spheres = env_knob("PT_SAH_MAX_SPHERES"); k.beams = env_knob("PT_BEAMS"); k.wide = env_knob("PT_WIDE"); k.descent = env_knob("PT_DESCENT"); k.roctx = env_knob("PT_ROCTX");
    return k;
}

EventPair* next_events(PtContext* c, int kind)
{
    if (c->ev_used >= (1u << 20)) return nullptr;  // bounded pool: profiling left on for a long run simply stops recording
    if (c->ev_used == c->ev_pool.size()) {
        EventPair p{};
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        c->ev_pool.push_back(p);
    }
    EventPair* p = &c->ev_pool[c->ev_used++];
    p->kind = kind;
    return p;
}

// DESIGN.md byte model (what the wavefront formulation must move through HBM).
//   fused schedule: secondary ray = 48 (written by the pass that spawns it) + 48 (read by the pass that traces it) = 96 B;
//                   primaries are generated, traced and shaded in registers: 0 B; pixel = 16 B final store
//   split schedule: secondary ray = 48 + 32 + 8 (traverse reads o,d, writes the hit) + 56 (shade reads ray + hit) = 144 B;
//                   primary slot = 56 (primary writes ray + hit) + 56 (shade reads them) = 112 B; pixel = 16 B
//   spp > 1 (both): radiance read-modify-write (32) per sample + primary-hit cache (8 write + 8 read) = 48 B per path
uint64_t bytes_per_secondary(bool split) { return split ? 144ull : 96ull; }
uint64_t fixed_bytes(bool split, uint64_t slots, uint64_t pixels, uint64_t spp_paths)
{
    return (split ? 112ull * slots : 0ull) + 16ull * pixels + 48ull * spp_paths;
}

FrameCounters make_counters(const Lane& L, uint32_t parity)
{
    FrameCounters fc{};
    fc.counts = L.d_counts + (size_t)parity * 2 * L.cap_counts;  // [queue sizes | work cursors]
    fc.fold_counts = L.d_counts + (size_t)(parity ^ 1u) * 2 * L.cap_counts;
    fc.n_counts = (uint32_t)L.cap_counts - 1u;  // the whole allocation is summed / zeroed when folded
    fc.tail_rays = L.d_totals + 2 + parity;
    fc.fold_tail = L.d_totals + 2 + (parity ^ 1u);
    fc.totals = L.d_totals;
    fc.host_counts = L.d_prev_counts;
    return fc;
}

// fold both parities of a lane into its totals (leaves all per-frame counters zero): the older frame first, so that the
// host-mapped queue sizes end up describing the lane's latest frame (counters that were already folded publish nothing)
hipError_t flush_all_counters(Lane& L)
{
    if (!L.d_counts) return hipSuccess;
    for (uint32_t i = 0; i < 2; i++) {
        const FrameCounters fc = make_counters(L, L.parity ^ 1u ^ i);
        hipError_t e = launch_flush_counters(fc.counts, fc.n_counts, fc.tail_rays, fc.totals, fc.host_counts, L.stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void sum_events(PtContext* c, size_t begin, size_t end, PtStats* stats)
{
    for (size_t i = begin; i < end; i++) {
        float t = 0;
        if (hipEventElapsedTime(&t, c->ev_pool[i].a, c->ev_pool[i].b) != hipSuccess) continue;
        switch (c->ev_pool[i].kind) {
            case 2: stats->ms_shade += t; stats->shade_launches++; break;
            case 3: stats->ms_tail += t; stats->tail_launches++; break;
            default: stats->ms_traverse += t; stats->traverse_launches++; break;  // primary / fused bounce / traverse
        }
    }
}

// Launches the build of primary-beam lists into `dst` (centre, slack: Beam), on stream `on` or, if null, on the cache's side stream.
static PtStatus beam_build(PtContext* c, const PixelMap& pm, PtContext::BeamLists* dst, const float centre[3], float slack, std::vector<uint32_t> key, hipStream_t on)
{
    auto& B = c->beam;
    if (!B.ev_last_use) PT_HIP(c, hipEventCreateWithFlags(&B.ev_last_use, hipEventDisableTiming));
    if (!on) {
        if (!B.stream) PT_HIP(c, hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
        on = B.stream;
    }
    if (!dst->ev_ready) PT_HIP(c, hipEventCreateWithFlags(&dst->ev_ready, hipEventDisableTiming));
    const size_t n_blocks = pm.n_slots >> 6;
    if (n_blocks > dst->cap_blocks) {
        PT_HIP(c, sync_all(c));  // frames in flight may read the old lists
        if (B.stream) PT_HIP(c, hipStreamSynchronize(B.stream));
        free_dev(dst->d_lists);
        dst->cap_blocks = 0;
        PT_HIP(c, hipMalloc(&dst->d_lists, n_blocks * 16u * sizeof(uint32_t)));
        dst->cap_blocks = n_blocks;
        dst->used = false;
    }
    // an earlier build into this buffer may still run (on another lane's stream), and the frames in flight may still read the buffer's
    // previous lists: the build waits for both (device-side waits only)
    if (dst->building) PT_HIP(c, hipStreamWaitEvent(on, dst->ev_ready, 0));
    if (dst->used)
        for (uint32_t i = 0; i < c->n_lanes; i++) {
            if (c->lanes[i].stream == on) continue;  // (in order behind them anyway)
            PT_HIP(c, hipEventRecord(B.ev_last_use, c->lanes[i].stream));
            PT_HIP(c, hipStreamWaitEvent(on, B.ev_last_use, 0));
        }
    FrameParams fp = make_frame_params(c);
    fp.cam.Position = make_f3(centre[0], centre[1], centre[2]);
    PT_HIP(c, launch_beams(make_scene_view(c), pm, fp, slack, dst->d_lists, on));
    PT_HIP(c, hipEventRecord(dst->ev_ready, on));
    dst->key = std::move(key);
    std::memcpy(dst->pos, centre, 12);
    dst->slack = slack;
    dst->building = true;
    dst->used = false;
    return PT_OK;  // this frame still traverses (or uses the lists it has); later ones find the new lists
}

// Primary-beam cache (PtContext::BeamCache).  *lists = lists that hold for this frame's view, else null; *wait = an event the frame must
// wait for before it reads them (the first frames of a resting view), or null.
PtStatus beam_cache_lookup(PtContext* c, const PixelMap& pm, uint32_t max_job_blocks, const uint32_t** lists, hipEvent_t* wait, BeamJob* job)
{
    *lists = nullptr;
    *wait = nullptr;
    *job = BeamJob{};
    auto& B = c->beam;
    // What the lists of a view have in common whatever the camera's pose: the frame geometry, the scene (the key) and the lens -- the lengths of
    // the camera's axes, which a turning camera reproduces only to rounding: compared in pixels (lens_px), like the turn itself.
    auto len3 = [](const float* v) { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); };
    const double len_r = len3(c->cam.RightDirection), len_u = len3(c->cam.UpDirection), len_f = len3(c->cam.ForwardDirection);
    std::vector<uint32_t> key;
    key.reserve(24);
    for (uint32_t v : { pm.mode, pm.img_w, pm.img_h, pm.rx, pm.ry, pm.rw, pm.rh, pm.ts, pm.first, pm.run, pm.stride, pm.n_slots,
                        (uint32_t)c->scene_gen, (uint32_t)(c->scene_gen >> 32) }) key.push_back(v);
    const float* pos = c->cam.Position;
    float basis[9];  // this frame's orientation: Right, Up, Forward as the kernels use them
    std::memcpy(basis, c->cam.RightDirection, 12); std::memcpy(basis + 3, c->cam.UpDirection, 12); std::memcpy(basis + 6, c->cam.ForwardDirection, 12);
    bool finite_pose = std::isfinite(pos[0]) && std::isfinite(pos[1]) && std::isfinite(pos[2]) && len_r > 0.0 && len_u > 0.0 && len_f > 0.0;
    for (float x : basis) finite_pose = finite_pose && std::isfinite(x);
    auto dist = [](const float* a, const float* b) {
        const double dx = (double)a[0] - b[0], dy = (double)a[1] - b[1], dz = (double)a[2] - b[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    // The rotation that takes orientation p to orientation q (both with this lens) as a rotation vector (axis * angle): R = Q * P^T over the
    // normalised axes; angle from the trace, axis from the antisymmetric part.
    auto rotation_between = [&](const float* p, const float* q, double w[3]) {
        double R[3][3] = {};
        for (int k = 0; k < 3; k++) {
            const double lp = len3(p + 3 * k), lq = len3(q + 3 * k), inv = lp > 0.0 && lq > 0.0 ? 1.0 / (lp * lq) : 0.0;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R[i][j] += (double)q[3 * k + i] * (double)p[3 * k + j] * inv;
        }
        const double ax[3] = { R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1] };  // 2 sin(angle) * axis
        const double s2 = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]), c2 = R[0][0] + R[1][1] + R[2][2] - 1.0;  // 2 sin, 2 cos
        const double angle = std::atan2(s2, c2);
        for (int i = 0; i < 3; i++) w[i] = s2 > 0.0 ? ax[i] / s2 * angle : 0.0;
        return angle;  // in [0, pi]; a half turn has no axis here, and nothing below builds lists for one
    };
    // How far, in pixels, a ray's crossing of the image can move when the camera turns by `angle` (any axis): a direction moves by at most that
    // angle, and at an angle a off the view axis a change of direction moves the crossing by at most f / cos^2(a) pixels per radian (f = the focal
    // length in pixels) -- taken at the image corner, plus the turn itself.
    const double fx_px = 0.5 * (double)pm.img_w * len_f / len_r, fy_px = 0.5 * (double)pm.img_h * len_f / len_u;  // (equal for square pixels)
    const double f_px = std::max(fx_px, fy_px);
    const double corner = std::atan(std::sqrt(len_r * len_r + len_u * len_u) / len_f);  // the image corner's angle off the view axis
    auto turn_px = [&](double angle) { const double a = std::min(corner + angle, 1.55), cs = std::cos(a); return angle * f_px / (cs * cs) * 1.01; };
    // ... and when its axes' lengths differ (by rounding: relative differences times the image's half diagonal, with the tangent at the corner)
    const double half_diag_px = 0.5 * std::sqrt((double)pm.img_w * pm.img_w + (double)pm.img_h * pm.img_h);
    auto lens_px = [&](const float* p, const float* q) {
        double worst = 0.0;
        for (int k = 0; k < 3; k++) { const double lp = len3(p + 3 * k), lq = len3(q + 3 * k); worst = std::max(worst, lp > 0.0 && lq > 0.0 ? std::fabs(lq / lp - 1.0) : 1e30); }
        const double cs = std::cos(std::min(corner, 1.55));
        return 2.0 * worst * half_diag_px / (cs * cs) * 1.01;
    };
    const uint64_t call = ++B.calls;
    auto within = [&](const PtContext::BeamLists& b, const float* q_pos, const float* q_basis) {
        if (b.slack == 0.0f ? std::memcmp(b.pos, q_pos, 12) != 0 : dist(b.pos, q_pos) > (double)b.slack * (1.0 - 1e-4)) return false;
        if (std::memcmp(b.basis, q_basis, 36) == 0) return true;
        if (b.margin_px == 0.0f) return false;
        double w[3];
        return turn_px(rotation_between(b.basis, q_basis, w)) + lens_px(b.basis, q_basis) <= (double)b.margin_px * (1.0 - 1e-3);
    };
    auto holds = [&](const PtContext::BeamLists& b) { return !b.key.empty() && b.key == key && call >= b.first_call && within(b, pos, basis); };
    const bool same_view = finite_pose && key == B.last_key && lens_px(B.last_basis, basis) <= 0.02;  // (the pose may differ)
    const double step = same_view ? dist(pos, B.last_pos) : 0.0;  // the camera's travel since the previous call ...
    double turn[3] = { 0.0, 0.0, 0.0 };
    const double turned = same_view ? rotation_between(B.last_basis, basis, turn) : 0.0;  // ... and its turn
    const bool rested = same_view && std::memcmp(B.last_pos, pos, 12) == 0 && std::memcmp(B.last_basis, basis, 36) == 0;
    const float prev[3] = { B.last_pos[0], B.last_pos[1], B.last_pos[2] };
    B.last_key = key;
    std::memcpy(B.last_pos, pos, 12);
    std::memcpy(B.last_basis, basis, 36);
    PtContext::BeamLists* cur = &B.buf[B.cur];
    PtContext::BeamLists* nxt = &B.buf[B.cur ^ 1];
    for (auto& b : B.buf)
        if (b.building) {
            if (hipEventQuery(b.ev_ready) == hipSuccess) b.building = false;
            else (void)hipGetLastError();
        }
    // Lists in use first; else the other buffer's, if they hold (the newer build).  A build the host has not seen finish is waited for ON THE
    // DEVICE: the host runs many frames ahead of the GPU, so "finished" at call time means little -- what counts is the order of the streams.
    // (the other buffer's lists also when they are the newer ones and no frame would have to wait for them: a moving camera's next lists are
    // centred further ahead, a camera that has stopped gets exact ones)
    if (holds(*nxt) && (!holds(*cur) || (nxt->built_call > cur->built_call && !nxt->building))) {
        B.cur ^= 1;
        std::swap(cur, nxt);
    }
    if (holds(*cur)) {
        *lists = cur->d_lists;
        cur->used = true;
        cur->last_use_call = call;
        if (cur->building) *wait = cur->ev_ready;
    }
    // The camera's velocities and how much they changed since the call before (for the extrapolation below)
    const double v[3] = { (double)pos[0] - prev[0], (double)pos[1] - prev[1], (double)pos[2] - prev[2] };
    double acc = 0.0, turn_acc = 0.0;
    if (same_view && B.have_vel) {
        const double d0 = v[0] - B.last_vel[0], d1 = v[1] - B.last_vel[1], d2 = v[2] - B.last_vel[2];
        acc = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        const double t0 = turn[0] - B.last_turn[0], t1 = turn[1] - B.last_turn[1], t2 = turn[2] - B.last_turn[2];
        turn_acc = std::sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    }
    B.have_vel = same_view;
    for (int i = 0; i < 3; i++) { B.last_vel[i] = (float)v[i]; B.last_turn[i] = turn[i]; }
    // A build in shares: this frame carries the next one -- unless the view it was planned for is gone (another lens, rect or scene, a stop)
    if (B.inc.active && (B.inc.key != key || rested || !finite_pose)) { B.inc.active = false; }
    auto make_job = [&](uint32_t first, uint32_t n) {
        const auto& I = B.inc;
        BeamJob j{};
        j.lists = I.dst->d_lists; j.first_block = first; j.n_blocks = n; j.slack = I.slack; j.margin_px = I.margin_px;
        std::memcpy(j.centre, I.centre, 12);
        std::memcpy(j.right, I.basis, 12); std::memcpy(j.up, I.basis + 3, 12); std::memcpy(j.forward, I.basis + 6, 12);
        return j;
    };
    auto finish = [&](PtContext::BeamLists* d) {
        // complete with this frame.  No event: a frame starts only after the frames n_lanes and more calls before it have finished
        // (render_common's marker on the caller's stream), so the frames from n_lanes calls on may read the lists.
        const auto& I = B.inc;
        d->key = key;
        std::memcpy(d->pos, I.centre, 12);
        std::memcpy(d->basis, I.basis, 36);
        d->slack = I.slack;
        d->margin_px = I.margin_px;
        d->first_call = call + c->n_lanes;
        d->building = false;
        d->built_call = call;
    };
    if (B.inc.active) {
        auto& I = B.inc;
        const uint32_t n = std::min(max_job_blocks, I.n_blocks - I.next_block);
        *job = make_job(I.next_block, n);
        I.next_block += n;
        I.dst->used = true;            // (a frame in flight WRITES the buffer: whatever builds into it next waits for the lanes, like after a reader)
        I.dst->last_use_call = call;
        if (I.next_block >= I.n_blocks) { I.active = false; finish(I.dst); }
        return PT_OK;
    }
    // Start a build?  Only for a view (lens, geometry, scene) that has lasted two calls, into the buffer frames do not use.
    if (!same_view) return PT_OK;
    // Where a build goes: never over the newest lists of this view (they may not be readable yet -- first_call -- but they are the future)
    auto of_view = [&](const PtContext::BeamLists& b) { return !b.key.empty() && b.key == key; };
    PtContext::BeamLists* dst = *lists ? nxt : cur;  // nothing usable in `cur`: build there (no swap needed later) ...
    if (!*lists && of_view(*cur) && (!of_view(*nxt) || cur->built_call > nxt->built_call)) dst = nxt;  // ... unless `cur` holds what is about to become usable
    if (rested) {
        if (*lists && cur->slack == 0.0f && cur->margin_px == 0.0f) return PT_OK;  // resting, and served by exact lists (a camera that has just stopped swaps its widened ones for exact ones)
        if (holds(*nxt) && nxt->slack == 0.0f && nxt->margin_px == 0.0f) return PT_OK;  // (already being built)
        dst->first_call = call + 1;
        dst->built_call = call;
        std::memcpy(dst->basis, basis, 36);
        dst->margin_px = 0.0f;
        return beam_build(c, pm, dst, pos, 0.0f, std::move(key), nullptr);  // a resting view's: one launch on the side stream; the next frame waits for it, once
    }
    // Moving.  The lists are built in shares of max_job_blocks blocks inside the primary passes of the next n_build frames (a build as a launch
    // of its own, behind a frame on its lane, cost that lane 60 us and the three lanes their even spacing for several frames: 0.083 ms per frame
    // where the lists themselves would give 0.078), are readable n_lanes calls after the last share, and are made for the `reach` frames from
    // then on: centred on the position -- and turned to the orientation -- extrapolated to the middle of that span, with half the span's travel,
    // two frames' and what the velocity's last change would add up to over the extrapolation as slack, and the same of the turn, in pixels, as
    // margin.  The next build starts when the lists in use will have run out by the time it could be ready.
    const uint32_t reach = knob_or(c->knobs.beam_reach, 32u);
    if (reach == 0u || max_job_blocks == 0u) return PT_OK;  // (PT_BEAM_REACH=0: lists for resting views only)
    const size_t n_blocks = pm.n_slots >> 6;
    if (n_blocks == 0) return PT_OK;
    const double n_build = (double)((n_blocks + max_job_blocks - 1) / max_job_blocks), a = (double)c->n_lanes;
    auto ahead = [&](double f, float out_pos[3], float out_basis[9]) {
        for (int i = 0; i < 3; i++) out_pos[i] = (float)((double)pos[i] + f * v[i]);
        // this frame's axes turned by f times the last turn (Rodrigues), their lengths kept
        const double ang = f * turned;
        std::memcpy(out_basis, basis, 36);
        if (ang > 0.0 && turned > 0.0) {
            const double k[3] = { turn[0] / turned, turn[1] / turned, turn[2] / turned }, cs = std::cos(ang), sn = std::sin(ang);
            for (int x = 0; x < 3; x++) {
                const double p[3] = { basis[3 * x], basis[3 * x + 1], basis[3 * x + 2] };
                const double kxp[3] = { k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0] }, kp = k[0] * p[0] + k[1] * p[1] + k[2] * p[2];
                for (int i = 0; i < 3; i++) out_basis[3 * x + i] = (float)(p[i] * cs + kxp[i] * sn + k[i] * kp * (1.0 - cs));
            }
        }
    };
    float then_pos[3], then_basis[9];
    ahead(n_build + a + 1.0, then_pos, then_basis);
    auto covers = [&](const PtContext::BeamLists& b) { return !b.key.empty() && b.key == key && (b.slack > 0.0f || b.margin_px > 0.0f) && within(b, then_pos, then_basis); };
    // (one build straight after the other was measured too: 200 of 300 frames find lists instead of 281 -- all but the first build's 19 -- because
    // a buffer must rest n_lanes calls between its last reader and its next build: 0.0815 against 0.0795 ms)
    if (covers(*cur) || covers(*nxt)) return PT_OK;
    // frames in flight may still read the buffer's previous lists: they were handed out no later than last_use_call, and a frame starts only
    // after the frames n_lanes and more calls before it have finished
    if (dst->last_use_call != 0 && call < dst->last_use_call + c->n_lanes) return PT_OK;
    if (dst->building) return PT_OK;  // (a resting view's build on the side stream still writes it)
    // The slack widens every pyramid by an absolute distance: at most the smallest radius (wider lists cost little -- 0.3 % of a frame at half
    // that -- but every block overflows in the end); the margin by pixels: at most PT_BEAM_MAX_MARGIN (8).  A faster camera gets lists for
    // fewer frames; one that jumps or spins gets none until it settles.
    double span = (double)reach;
    const double max_slack = (double)c->min_radius * 0.01 * (double)knob_or(c->knobs.beam_max_slack_pct, 100u), max_margin = (double)knob_or(c->knobs.beam_max_margin, 8u);
    auto slack_for = [&](double sp) { const double t = n_build + a + 0.5 * sp; return (0.5 * sp + 2.0) * step + 0.75 * acc * t * t; };
    auto margin_for = [&](double sp) { const double t = n_build + a + 0.5 * sp; return turned > 0.0 || turn_acc > 0.0 ? turn_px((0.5 * sp + 2.0) * turned + 0.75 * turn_acc * t * t) + 0.05 : 0.0; };
    while (span >= 4.0 && (slack_for(span) > max_slack || margin_for(span) > max_margin)) span -= 2.0;
    if (!(span >= 4.0)) return PT_OK;
    const float slack = (float)slack_for(span), margin_px = (float)margin_for(span);
    float mid[3], mid_basis[9];
    ahead(n_build - 1.0 + a + 0.5 * span, mid, mid_basis);
    bool ok = (slack > 0.0f || margin_px > 0.0f) && std::isfinite(slack) && std::isfinite(margin_px);
    for (float x : mid) ok = ok && std::isfinite(x);
    for (float x : mid_basis) ok = ok && std::isfinite(x);
    if (!ok) return PT_OK;
    if (n_blocks > dst->cap_blocks) {
        PT_HIP(c, sync_all(c));  // frames in flight may read the old lists
        if (B.stream) PT_HIP(c, hipStreamSynchronize(B.stream));
        free_dev(dst->d_lists);
        dst->cap_blocks = 0;
        PT_HIP(c, hipMalloc(&dst->d_lists, n_blocks * 16u * sizeof(uint32_t)));
        dst->cap_blocks = n_blocks;
    }
    dst->key.clear();  // (nothing may take the buffer's old lists from here on)
    dst->used = true;   // (written by frames in flight from now on)
    dst->last_use_call = call;
    auto& I = B.inc;
    I.active = true; I.dst = dst; I.key = key; I.slack = slack; I.margin_px = margin_px; I.next_block = 0; I.n_blocks = (uint32_t)n_blocks;
    std::memcpy(I.centre, mid, 12);
    std::memcpy(I.basis, mid_basis, 36);
    const uint32_t n = std::min(max_job_blocks, I.n_blocks);
    *job = make_job(0u, n);
    I.next_block = n;
    if (I.next_block >= I.n_blocks) { I.active = false; finish(dst); }
    return PT_OK;
}

// Moving spheres (row N2).  Every lane owns a copy of what moves (spheres, Morton-ordered spheres, node boxes) and a pinned staging
// buffer.  stage_spheres_on_lane puts new spheres into L's staging buffer and gets them onto the device -- for small scenes by leaving
// them there for refit_lane's single kernel (upload_pending), else with a copy on the lane's stream; refit_lane recomputes L's boxes.
static PtStatus stage_spheres_on_lane(PtContext* c, Lane& L, const PtSphere* spheres)
{
    const uint32_t n = c->n;
    if (L.scene_n != n) {
        PT_HIP(c, hipStreamSynchronize(L.stream));
        free_lane_scene(L);
        PT_HIP(c, hipMalloc(&L.d_sph, (size_t)n * sizeof(float4)));
        PT_HIP(c, hipMalloc(&L.d_sph_sorted, (size_t)n * sizeof(float4)));
        PT_HIP(c, hipMalloc(&L.d_nodes, (size_t)std::max(1u, n - 1) * sizeof(PtBvhNode)));
        PT_HIP(c, hipMalloc(&L.d_refit_flags, (size_t)n * sizeof(uint32_t)));
        PT_HIP(c, hipMalloc(&L.d_refit_hdr, 16 * sizeof(uint32_t)));
        PT_HIP(c, hipHostMalloc(&L.h_stage, (size_t)n * sizeof(PtSphere), hipHostMallocMapped));
        { void* dp = nullptr; PT_HIP(c, hipHostGetDevicePointer(&dp, L.h_stage, 0)); L.d_stage = static_cast<const float4*>(dp); }
        if (!L.ev_upload) PT_HIP(c, hipEventCreateWithFlags(&L.ev_upload, hipEventDisableTiming));
        L.scene_n = n;
    }
    if (!L.scene_private) {
        // first update on this lane: start from the master tree (topology + boxes), ordered after its build
        PT_HIP(c, hipStreamSynchronize(c->stream));
        if (c->n_nodes) PT_HIP(c, hipMemcpyAsync(L.d_nodes, c->d_nodes, (size_t)c->n_nodes * sizeof(PtBvhNode), hipMemcpyDeviceToDevice, L.stream));
        L.scene_private = true;
    } else {
        PT_HIP(c, hipEventSynchronize(L.ev_upload));  // the previous upload from the staging buffer has been consumed
    }
    std::memcpy(L.h_stage, spheres, (size_t)n * sizeof(PtSphere));
    L.needs_refit = true;
    // Small scenes: no copy here -- refit_lane's single kernel reads the staging buffer (pinned host memory) itself, so a frame of
    // an animated scene starts with ONE launch instead of a copy and four launches (DESIGN row N2).
    if (lbvh_gpu_refit_fused_possible(n) && knob_or(c->knobs.fused_refit, 1u) != 0) { L.upload_pending = true; return PT_OK; }
    PT_HIP(c, hipMemcpyAsync(L.d_sph, L.h_stage, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, L.stream));
    PT_HIP(c, hipEventRecord(L.ev_upload, L.stream));
    return PT_OK;
}

static PtStatus refit_lane(PtContext* c, Lane& L)
{
    L.needs_refit = false;
    if (L.upload_pending) {
        L.upload_pending = false;
        PT_HIP(c, lbvh_gpu_refit_fused(c->gpu_builder, L.d_stage, L.d_sph, c->n, reinterpret_cast<PtBvhNode*>(L.d_nodes), L.d_sph_sorted,
                                       c->d_sorted_id, L.d_refit_hdr, L.stream));
        PT_HIP(c, hipEventRecord(L.ev_upload, L.stream));  // the staging buffer has been read once this kernel is done
        return PT_OK;
    }
    PT_HIP(c, lbvh_gpu_refit(c->gpu_builder, L.d_sph, c->n, reinterpret_cast<PtBvhNode*>(L.d_nodes), L.d_sph_sorted, c->d_sorted_id,
                             L.d_refit_flags, L.d_refit_hdr, c->depth, L.stream));
    return PT_OK;
}

// pt_update_spheres reaches the lane of the NEXT frame only; the frames after it run on other lanes, and they must see the moved spheres
// too (an application that moves its spheres once and then renders on -- found by tests/test_gpu_stateful.py: two of three frames showed
// the old positions).  A lane that is behind takes the newest spheres over from the staging buffer of the lane that received them and
// refits its own boxes, on its own stream, before it renders.  An animated scene updates before every frame and never comes here.
static PtStatus sync_lane_spheres(PtContext* c, Lane& L)
{
    if (c->latest_lane < 0 || L.sph_gen == c->sph_gen) return PT_OK;
    const Lane& src = c->lanes[c->latest_lane];
    if (PtStatus st = stage_spheres_on_lane(c, L, reinterpret_cast<const PtSphere*>(src.h_stage)); st != PT_OK) return st;
    L.sph_gen = c->sph_gen;
    return refit_lane(c, L);
}

// The per-frame launch sequence.  out: device float4 buffer addressed by PixelRef::out_index.
//
// Fused schedule (default):  bounce<primary> -> bounce (x S) -> bounce<loop>
//   every kernel traces its rays and runs one shade step; S compacting wavefront bounces, then the looping form
//   finishes every remaining path in one launch.
// Split schedule (PT_FLAG_SPLIT_KERNELS / PT_SPLIT=1):  primary -> shade(0) -> [traverse(k) -> shade(k)] x S -> tail
//   separate traverse and shade kernels with a hit stream in between.
// S = PT_TAIL_AFTER for spp == 1; with spp > 1 (sample regeneration keeps the queue full) the host polls the queue
// size after every pass and switches to the looping kernel when it drops below PT_TAIL_THRESHOLD rays.
PtStatus render_common(PtContext* c, const PixelMap& pm, uint64_t valid_pixels, float4* out, PtStats* stats)
{
    const RoctxRange range(c, pm.mode == 0 ? "pt_render" : "pt_render_tiles");
    const uint32_t bounces = c->gs.Bounces, spp = c->gs.SamplesPerPixel;
    if (const uint32_t env = c->sd.EnvironmentLightTextureDescriptor; env != ~0u) {
        const size_t n_faces = c->sd.IsEnvironmentLightTextureCubeMap ? 6 : 1;
        if (!c->has_textures || (size_t)env + n_faces > c->tex_dims.size())
            return fail(c, PT_ERR_STATE, "SceneData.EnvironmentLightTextureDescriptor names a texture the table of pt_set_textures does not hold (call pt_set_textures after pt_set_scene; a cube map takes six consecutive entries)");
        for (size_t f = 0; f < n_faces; f++)
            if (n_faces == 6 && (c->tex_dims[env + f].first != c->tex_dims[env].first || c->tex_dims[env + f].second != c->tex_dims[env].first))
                return fail(c, PT_ERR_STATE, "SceneData.EnvironmentLightTextureDescriptor: the six faces of a cube map must be square and of one size");
    }
    const size_t max_iters = (size_t)spp * bounces + 1;  // passes if everything ran as wavefront
    const size_t wf_cap = spp > 1 ? max_iters : std::min<size_t>(max_iters, 64);
    // LDS-resident BVH: fused trace+shade passes (traversal is cheap, the hit stream is pure overhead).  BVH in global
    // memory: separate traverse kernels (43 VGPRs, 8 waves/SIMD hide the node-fetch latency; the fused kernel only
    // reaches 4) -- measured 4.3 vs 5.3 ms per frame on the 2^20-sphere scene.
    const bool split = (c->flags & PT_FLAG_SPLIT_KERNELS) || knob_or(c->knobs.split, c->lds_scene ? 0u : 1u) != 0;
    // every check that can reject the frame comes before any state change (lane rotation, counter parity, markers)
    if (traverse_lds_bytes_for(c->n_nodes, c->n, stack_entries(c, c->d_wide != nullptr), c->lds_scene) > kMaxLdsBytes - 9u * 1024u)  // (the kernels' static LDS comes on top)
        return fail(c, PT_ERR_UNSUPPORTED, "BVH depth needs more traversal-stack LDS than a workgroup can have");
    // frames in flight: this frame runs on the next lane (its own stream and work buffers); the rotation itself happens
    // below, once the lane's buffers exist
    Lane& L = c->lanes[c->next_lane];
    if (wf_cap + 2 > L.cap_counts && L.cap_counts) {
        // growing the counter arrays: fold what the old ones hold into the totals first
        PT_HIP(c, flush_all_counters(L));
        PT_HIP(c, hipStreamSynchronize(L.stream));
    }
    const bool di = c->gs.IsDIEnabled && c->n_lights > 0;
    // Is the caller waiting for each frame (App::Tick -> Render -> WaitForGPU) or keeping several in flight?  Asked of the streams before this
    // frame queues anything: all lanes drained = one frame at a time, and the frame is scheduled for latency (below: the fused form).
    bool lanes_idle = true;
    for (uint32_t i = 0; i < c->n_lanes && lanes_idle; i++)
        if (hipStreamQuery(c->lanes[i].stream) != hipSuccess) { lanes_idle = false; (void)hipGetLastError(); }
    // Persistent workgroups: with the BVH staged into LDS per workgroup, 2 per CU (= the 4 waves/SIMD the kernel is built
    // for) amortise the 37 KB staging over ~4 batches of rays at 1080p / 1 spp (0.121 -> 0.116 ms per frame), from about
    // 1.5 M slots: below that 8 per CU is 4-10 % faster.  (Since the waves of a workgroup draw their tiles dynamically the
    // same holds at spp > 1 -- C3: 3.61 ms with 2 per CU, 3.68 with 8.)
    const bool big_frame = c->lds_scene && pm.n_slots >= 1500000u;
    const uint32_t trav_cap = c->num_cus * knob_or(c->knobs.traverse_blocks_per_cu, big_frame ? 2 : 8);
    const uint32_t fused_threads = c->lds_scene ? knob_or(c->knobs.fused_threads, 512) : 256u;
    auto grid_for = [](uint32_t items, uint32_t threads, uint32_t cap) { return std::max(1u, std::min((items + threads - 1) / threads, cap)); };
    // Segmented hand-over from the primary pass to the looping pass (FrameCounters::seg_counts): workgroup b owns the queue
    // entries [b * seg_cap, (b + 1) * seg_cap), seg_cap = the slots it visits -- the queues get that much room
    const uint32_t primary_grid = grid_for(pm.n_slots, fused_threads, trav_cap);
    const uint32_t primary_batches = (pm.n_slots + fused_threads - 1) / fused_threads;
    const uint32_t seg_cap = (primary_batches + primary_grid - 1) / primary_grid * fused_threads;
    const size_t seg_total = (size_t)primary_grid * seg_cap;
    const bool seg_possible = !split && max_iters > 1 && primary_grid <= kMaxSegs && knob_or(c->knobs.seg, 1u) != 0 && c->knobs.loop_use_tail < 0;
    PtStatus st = ensure_buffers(c, L, split ? pm.n_slots : std::max<size_t>(pm.n_slots, seg_total), spp > 1, split, wf_cap + 2, di);
    if (st != PT_OK) return st;
    if ((st = sync_lane_rotations(c, L)) != PT_OK) return st;
    if ((st = sync_lane_spheres(c, L)) != PT_OK) return st;
    if (L.needs_refit && (st = refit_lane(c, L)) != PT_OK) return st;  // pt_update_spheres without pt_refit_accel: the frame refits by itself
    c->last_lane = c->next_lane;
    c->next_lane = (c->next_lane + 1) % c->n_lanes;
    // Primary beams: use cached candidate lists that hold for this frame's view (beam_cache_lookup): exact ones for a view that has rested
    // for two frames, ones with slack for a camera that moves without turning.  PT_BEAMS=0 switches them off, PT_BEAM_REACH=0 the moving
    // kind, for A/B runs.  (Building lists in front of EVERY frame of a changing view was measured, with the tree staged in LDS for the
    // build: the animated C2 frame went from 0.099 to 0.117 ms -- the build lengthens the frame's dependent chain by more than the
    // primary pass gains.  Animated scenes get none.)
    const uint32_t* beam_lists = nullptr;
    BeamJob beam_job{};
    if (!split && !L.scene_private && c->n_nodes > 0 && knob_or(c->knobs.beams, 1u) != 0 && std::fabs(c->cam.Jitter[0]) <= 0.5f && std::fabs(c->cam.Jitter[1]) <= 0.5f) {
        hipEvent_t beam_wait = nullptr;
        // (a share of a build rides on the first wave of up to PT_BEAM_SHARE_WGS = 128 workgroups of the primary pass, 64 blocks each: four frames'
        // shares make a 1080p build -- 32 left a camera that travels AND turns without lists for a sixth of its frames, the whole build in one
        // frame costs that frame 1 %)
        if ((st = beam_cache_lookup(c, pm, std::min(primary_grid, knob_or(c->knobs.beam_share_wgs, 128u)) * 64u, &beam_lists, &beam_wait, &beam_job)) != PT_OK) return st;
        if (beam_wait) PT_HIP(c, hipStreamWaitEvent(L.stream, beam_wait, 0));
    } else {
        c->beam.last_key.clear();
    }
    if (L.stream != c->stream) {
        // N frames in flight.  The caller rotates over N output buffers, so this frame may start as soon as the consumer of
        // ITS buffer (queued on the caller's stream right after the render call N calls ago, i.e. before the call N-1 calls
        // ago) has run: wait for the marker recorded at the start of that call -- not for the frames in between, whose
        // completion waits were queued on the caller's stream after that marker.
        const uint64_t nl = c->n_lanes;
        PT_HIP(c, hipEventRecord(c->ev_in[c->calls % nl], c->stream));
        // (a stream wait costs the host ~8 us, a query ~1: when the GPU is ahead of the host -- small frames -- the marker has
        // usually completed already and the wait is skipped)
        const hipEvent_t marker = c->ev_in[c->calls >= nl - 1 ? (c->calls - (nl - 1)) % nl : 0];
        if (hipEventQuery(marker) != hipSuccess) {
            (void)hipGetLastError();  // hipErrorNotReady is not an error here
            PT_HIP(c, hipStreamWaitEvent(L.stream, marker, 0));
        }
        // The rotation rule above is the caller's side of the contract; it is also enforced: when `out` is a buffer one of the
        // other lanes wrote within the window, this frame waits for the marker recorded at the start of THIS call -- i.e. for
        // everything the caller has queued so far, which includes the wait for that earlier frame and whatever consumed it.
        // (Costs the overlap of the frames, never correctness.)
        for (uint32_t i = 0; i < c->n_lanes; i++)
            if (&c->lanes[i] != &L && c->lanes[i].last_out == out) {
                PT_HIP(c, hipStreamWaitEvent(L.stream, c->ev_in[c->calls % nl], 0));
                break;
            }
        L.last_out = out;
        c->calls++;
    }

    const SceneView sv = make_scene_view(c, &L);
    FrameParams fp = make_frame_params(c);
    fp.beam_lists = beam_lists;
    fp.beam_job = beam_job;
    L.parity ^= 1u;
    const FrameCounters fc = make_counters(L, L.parity);
    uint32_t* counts = fc.counts;

    // Launch grids: a kernel's queue size lives on the device; the host sizes the grid from the queue sizes an
    // earlier frame of the same configuration had (1.25x margin), falling back to the n_slots upper bound.
    const uint64_t signature = ((uint64_t)pm.n_slots << 32) ^ ((uint64_t)bounces << 20) ^ ((uint64_t)spp << 4) ^ pm.mode ^ ((uint64_t)c->n << 40) ^ (split ? 8u : 0u);
    const bool have_prev = spp == 1 && L.prev_signature == signature && L.h_prev_counts[0] == pm.n_slots && c->knobs.no_adaptive_grid < 0;
    auto estimate = [&](size_t k) -> uint32_t {
        if (!have_prev || k >= L.cap_counts) return pm.n_slots;
        const uint64_t e = (uint64_t)L.h_prev_counts[k] + L.h_prev_counts[k] / 4 + 64;
        return (uint32_t)std::min<uint64_t>(e, pm.n_slots);
    };
    const uint32_t trav_cap_wide = c->num_cus * 8u;
    const uint32_t shade_cap = c->num_cus * knob_or(c->knobs.shade_blocks_per_cu, 16);
    // the looping pass: small queues at 1 spp (256 threads, up to 8 workgroups per CU); at spp > 1 of the fused schedule it
    // carries the whole frame after the primary pass (every lane stays busy until its pixel has all its samples), as 2
    // persistent 512-thread workgroups per CU (C3: 6.7 -> 4.8 ms per frame against 33 queue passes + a small looping pass)
    const bool loop_is_main = spp > 1 && !split && c->knobs.tail_threshold < 0;
    const uint32_t tail_cap = c->num_cus * knob_or(c->knobs.tail_blocks_per_cu, loop_is_main ? 2 : 8);
    const uint32_t trav_threads = traverse_threads(c->lds_scene);
    // looping pass: 512-thread workgroups when it carries the frame (spp > 1) and, at 1 spp, for big frames (1080p: 0.0954 ->
    // 0.0887 ms, 4K: 0.329 -> 0.302; at 960x540 and below 256 threads are 6-7 % faster)
    const uint32_t loop_threads = knob_or(c->knobs.loop_threads, c->lds_scene && (loop_is_main || big_frame) ? 512u : 256u);
    // Queue-fed passes before the looping kernel (spp == 1).  Fused, large frames: the primary pass also traces the first bounce
    // in registers (bounce_kernel kInline2) and the looping kernel follows it directly -- two launches per frame: 4-10 % faster at
    // every frame size from 256x256 to 4K (PT_INLINE2_MIN_SLOTS switches it off below a slot count, for A/B runs).
    const bool inline2 = !split && spp == 1 && pm.n_slots >= knob_or(c->knobs.inline2_min_slots, 0u);
    // (split: 4 queue passes and 4 persistent workgroups per CU for the ray-replacement kernel, re-measured in round 2 after the bounded
    // descent made these scenes VALU-bound -- 5 000 spheres 0.469 -> 0.436 ms, 2^20 2.42 -> 2.31, 4 M 5.19 -> 3.91; 10^5: 0.880 -> 0.901)
    const size_t tail_after = knob_or(c->knobs.tail_after, split ? 4 : (inline2 ? 0 : 1));
    const uint32_t tail_threshold = knob_or(c->knobs.tail_threshold, 262144);  // queue size below which spp > 1 switches to it

    const bool timed = stats != nullptr;
    const size_t ev_begin = c->ev_used;  // this frame's slice of the per-launch event pool
    if (timed) PT_HIP(c, hipEventRecord(c->ev0, L.stream));

    auto bracket = [&](int kind, auto&& launch) -> hipError_t {
        EventPair* ev = c->profiling ? next_events(c, kind) : nullptr;
        if (ev) (void)hipEventRecord(ev->a, L.stream);
        hipError_t e = launch();
        if (ev) (void)hipEventRecord(ev->b, L.stream);
        return e;
    };
    // decide, after pass k has filled queue k+1, whether the looping kernel takes over (and whether anything is left)
    auto poll = [&](size_t k, bool& empty, bool& go_loop) -> PtStatus {
        empty = false;
        if (loop_is_main) {
            go_loop = true;  // fused, spp > 1: no queue passes at all, hence nothing to poll
        } else if (spp > 1) {
            // Lagged polling: read back the size of queue k+1 asynchronously, but decide on the size the queue had kPollLag
            // passes ago, whose copy has long completed -- the GPU never idles waiting for the host.  Queue sizes only
            // shrink (a path emits at most one ray per pass), so a stale size is an upper bound: "was already empty" and
            // "was already below the threshold" stay true; the cost of staleness is at most kPollLag cheap extra passes.
            constexpr size_t kPollLag = 2;
            PT_HIP(c, hipMemcpyAsync(L.h_counts + k + 1, counts + k + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, L.stream));
            PT_HIP(c, hipEventRecord(L.ev_poll[k & 3], L.stream));
            go_loop = k + 2 >= wf_cap;
            if (k >= kPollLag) {
                PT_HIP(c, hipEventSynchronize(L.ev_poll[(k - kPollLag) & 3]));
                const uint32_t v = L.h_counts[k - kPollLag + 1];
                empty = v == 0;
                go_loop = go_loop || v < tail_threshold;
            }
        } else {
            go_loop = k >= tail_after || k + 2 >= wf_cap;
        }
        return PT_OK;
    };

    // The launch sequence.  A HIP failure in the middle of it must not leave the caller's stream unordered with respect to the
    // kernels that were launched: the completion event is recorded and waited for either way (the counter parity stays
    // flipped -- the kernels that did run used this parity's counters, and the lane's next frame folds them).
    auto submit = [&]() -> PtStatus {
        if (fp.di_enabled && split) {
            // row N4, split schedule: the direct-illumination estimate of every primary surface, before the shade passes read it
            // (the fused schedule's primary pass makes the estimate itself, at the first shading of the primary surface)
            const uint32_t di_grid = grid_for(pm.n_slots, trav_threads, trav_cap_wide);
            PT_HIP(c, bracket(1, [&] { return launch_di(sv, pm, fp, L.scratch.di, L.scratch.primary_hit, fc.tail_rays, di_grid, L.stream); }));
        }
        if (!split) {
            // The looping pass follows the primary pass directly (1 spp with the in-register second bounce; spp > 1): the hand-over
            // is segmented -- no barrier, no global atomic per batch in the primary pass (PT_SEG=0: the dense queue, for A/B runs).
            const bool loop_follows_primary = loop_is_main || (spp == 1 && (tail_after == 0 || wf_cap <= 2));
            const bool seg = loop_follows_primary && seg_possible;
            FrameCounters fc_seg = fc;
            // Small frames are bound by the dependent chain of a frame's launches, not by throughput: there the primary pass also
            // finishes the paths of its own segments and no looping pass is launched -- ONE launch per frame (256x256: 0.0194 ->
            // 0.0162 ms, 640x384: 0.0397 -> 0.0323, 640x384 at 4 spp: 0.131 -> 0.107, a 1/8 share of a 1080p frame in tiles: 0.0566 ->
            // 0.0516; a single 1080p frame in flight: 0.190 -> 0.159).  From about half a million slots, with several frames in
            // flight, the separate looping pass wins (960x540: 0.0426 vs 0.0491 fused, 720p: 0.058 vs 0.076, C2: 0.089 vs 0.096,
            // C3: 3.60 vs 3.69).  A 1-spp frame submitted to an idle context is such a lone frame whatever its size: fused, C2's one frame at a time
            // takes 0.171 instead of 0.193 ms (and 0.088 instead of 0.078 with three in flight, which is why the streams are asked).
            // PT_FUSE_LOOP=0/1 overrides.
            const bool fuse = seg && knob_or(c->knobs.fuse_loop, (pm.n_slots < 400000u || (lanes_idle && spp == 1)) ? 1u : 0u) != 0;
            if (seg) { fc_seg.seg_counts = L.d_seg_counts; fc_seg.n_segs = primary_grid; fc_seg.seg_cap = seg_cap; fc_seg.fuse_loop = fuse ? 1u : 0u; }
            // spp > 1 with a separate looping pass over an untextured scene: the primary pass leaves each pixel's primary-hit record for the
            // samples the looping pass regenerates, and where it writes the pixel (Scratch::primary_cache)
            Scratch scratch = L.scratch;
            if (sv.tex_maps || fuse || spp == 1) scratch.primary_cache = nullptr;  // (the untextured looping kernel of spp > 1 frames relies on the records)
            // pass 0 generates + traces the primaries and shades them into queue 1; pass k >= 1 consumes queue k
            for (size_t k = 0;; k++) {
                const RayQueue& qin = L.q[k & 1];
                const RayQueue& qout = L.q[(k + 1) & 1];
                const bool primary = k == 0;
                bool go_loop = false, empty = false;
                if (k > 0) {
                    // queue k exists: is it worth another compacting pass?
                    if ((st = poll(k - 1, empty, go_loop)) != PT_OK) return st;
                    if (empty) break;
                }
                const bool last_possible = k + 1 >= max_iters;  // no path can have another ray after this pass
                const bool loop = (k > 0 && go_loop);
                const uint32_t threads = loop ? loop_threads : fused_threads;
                const uint32_t items = primary ? pm.n_slots : estimate(k);
                const uint32_t cap = loop ? tail_cap : trav_cap;
                if (loop && c->knobs.loop_use_tail >= 0) {
                    PT_HIP(c, launch_tail(sv, pm, fp, qin, L.scratch, out, counts + k, fc.tail_rays, fc.totals, grid_for(items, kTailThreads, tail_cap), L.stream));
                    break;
                }
                PT_HIP(c, bracket(loop ? 3 : (primary ? 0 : 1), [&] {
                    return launch_bounce(sv, pm, fp, qin, qout, scratch, out, counts + k, counts + k + 1, k <= 1 ? fc_seg : fc, primary, loop, inline2, threads,
                                         primary ? primary_grid : grid_for(items, threads, cap), L.stream);
                }));
                if (loop || last_possible || (primary && fuse)) break;
            }
        } else {
            const uint32_t trav_grid = grid_for(pm.n_slots, trav_threads, trav_cap);
            PT_HIP(c, bracket(0, [&] { return launch_primary(sv, pm, fp, L.q[0], L.scratch, out, fc, trav_grid, L.stream); }));
            // shade pass k consumes queue k (counts[k]) and appends to queue k+1; queue k+1 is then traversed, or handed to the tail
            for (size_t k = 0;; k++) {
                const RayQueue& qin = L.q[k & 1];
                const RayQueue& qout = L.q[(k + 1) & 1];
                PT_HIP(c, bracket(2, [&] { return launch_shade(sv, pm, fp, qin, qout, L.scratch, out, counts + k, counts + k + 1, grid_for(estimate(k), kShadeThreads, shade_cap), L.stream); }));
                if (k + 1 == max_iters) break;  // no path can have another ray
                bool go_loop = false, empty = false;
                if ((st = poll(k, empty, go_loop)) != PT_OK) return st;
                if (empty) break;
                if (go_loop) {
                    PT_HIP(c, bracket(3, [&] { return launch_tail(sv, pm, fp, qout, L.scratch, out, counts + k + 1, fc.tail_rays, fc.totals, grid_for(estimate(k + 1), kTailThreads, tail_cap), L.stream); }));
                    break;
                }
                if (!c->lds_scene && sv.n > 1 && knob_or(c->knobs.ray_replacement, 1)) {
                    // persistent waves with ray replacement (heavy-tailed visit counts of large scenes)
                    uint32_t* cursor = counts + L.cap_counts + k + 1;
                    const uint32_t grid = std::min(grid_for(estimate(k + 1), 256u, c->num_cus * 8u), c->num_cus * knob_or(c->knobs.dyn_blocks_per_cu, 4));
                    PT_HIP(c, bracket(1, [&] { return launch_traverse_dyn(sv, qout, counts + k + 1, cursor, fc.totals, grid, L.stream); }));
                } else {
                    PT_HIP(c, bracket(1, [&] { return launch_traverse(sv, qout, counts + k + 1, grid_for(estimate(k + 1), trav_threads, trav_cap), L.stream); }));
                }
            }
        }
        return PT_OK;
    };
    st = submit();
    if (st != PT_OK && beam_job.n_blocks) {
        // the share this frame was to carry is lost: the lists it belongs to must never be taken
        c->beam.inc.active = false;
        for (auto& b : c->beam.buf) if (b.d_lists == beam_job.lists) b.key.clear();
    }
    // this frame's queue sizes reach h_prev_counts when the lane's next frame folds them (frame_counters_begin): no copy call
    L.prev_signature = signature;  // (the launch-grid estimates above only use them at 1 spp; pt_get_queue_sizes reports them always)
    if (L.stream != c->stream) {
        // whatever the caller queues next on its stream (a gather, a copy, the next use of `out`) sees the finished frame
        const hipError_t e1 = hipEventRecord(L.ev_done, L.stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(c->stream, L.ev_done, 0) : e1;
        if (st == PT_OK && e2 != hipSuccess) st = fail(c, PT_ERR_HIP, std::string("render: completion event: ") + hipGetErrorString(e2));
    }
    if (st != PT_OK) return st;
    c->tot_pixels += valid_pixels;
    c->tot_paths += valid_pixels * spp;
    c->tot_fixed_bytes += fixed_bytes(split, pm.n_slots, valid_pixels, spp > 1 ? valid_pixels * spp : 0);
    c->tot_sec_coeff = bytes_per_secondary(split);
    if (beam_lists) c->tot_beam_frames++;
    if (timed) {
        PT_HIP(c, hipEventRecord(c->ev1, L.stream));
        if (c->knobs.debug_counts >= 0) {
            std::vector<uint32_t> hc(L.cap_counts);
            PT_HIP(c, hipMemcpyAsync(hc.data(), counts, L.cap_counts * sizeof(uint32_t), hipMemcpyDeviceToHost, L.stream));
            PT_HIP(c, hipStreamSynchronize(L.stream));
            std::fprintf(stderr, "[pt] queue sizes:");
            for (size_t k = 0; k < L.cap_counts && k < 20; k++) std::fprintf(stderr, " %u", hc[k]);
            std::fprintf(stderr, "\n");
        }
        // fold the previous frame first (normally done by the next frame's first kernel), then this one, so that
        // totals[1] is this frame's secondary-ray count
        {
            const FrameCounters other = make_counters(L, L.parity ^ 1u);
            PT_HIP(c, launch_flush_counters(other.counts, other.n_counts, other.tail_rays, other.totals, other.host_counts, L.stream));
            PT_HIP(c, launch_flush_counters(fc.counts, fc.n_counts, fc.tail_rays, fc.totals, fc.host_counts, L.stream));
        }
        unsigned long long secondary = 0;
        PT_HIP(c, hipMemcpyAsync(&secondary, L.d_totals + 1, sizeof secondary, hipMemcpyDeviceToHost, L.stream));
        PT_HIP(c, hipStreamSynchronize(L.stream));
        std::memset(stats, 0, sizeof *stats);
        float ms = 0;
        PT_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        stats->ms_total = ms;
        stats->rays = valid_pixels + secondary;
        stats->pixels = valid_pixels;
        stats->paths = valid_pixels * spp;
        stats->bytes_algorithmic = bytes_per_secondary(split) * secondary + fixed_bytes(split, pm.n_slots, valid_pixels, spp > 1 ? valid_pixels * spp : 0);
        stats->beams_used = beam_lists ? 1u : 0u;
        if (c->profiling) sum_events(c, ev_begin, c->ev_used, stats);
    }
    return PT_OK;
}

// tiles t in [0, total) with first <= t % stride < first + run
uint32_t count_tiles(uint32_t total, uint32_t first, uint32_t run, uint32_t stride)
{
    const uint32_t rem = total % stride;
    return (total / stride) * run + (rem > first ? std::min(rem - first, run) : 0u);
}

uint64_t count_tile_pixels(uint32_t w, uint32_t h, uint32_t ts, uint32_t first, uint32_t run, uint32_t stride)
{
    const uint32_t tx = (w + ts - 1) / ts, ty = (h + ts - 1) / ts, total = tx * ty;
    uint64_t px = 0;
    for (uint32_t base = 0; base < total; base += stride)
        for (uint32_t t = base + first; t < std::min(base + first + run, total); t++) {
            const uint32_t x0 = (t % tx) * ts, y0 = (t / tx) * ts;
            px += (uint64_t)std::min(ts, w - x0) * std::min(ts, h - y0);
        }
    return px;
}

uint32_t frame_tiles(const PtContext* c)
{
    const uint32_t ts = c->tile_size;
    return ((c->gs.RenderSize[0] + ts - 1) / ts) * ((c->gs.RenderSize[1] + ts - 1) / ts);
}

}  // namespace

// ==================================================================================================== C-ABI
extern "C" {

const char* pt_version(void) { return "dxrs-amd 0.1 (gfx950)"; }

PtStatus pt_create(const PtConfig* config, PtContext** out_ctx)
{
    if (!config || !out_ctx) return PT_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return PT_ERR_NO_DEVICE;
    if (config->device < 0 || config->device >= count) return PT_ERR_INVALID_ARG;
    PtContext* c = new (std::nothrow) PtContext();
    if (!c) return PT_ERR_OOM;
    c->device = config->device;
    c->flags = config->flags;
    c->knobs = read_knobs();
    c->tile_size = config->tile_size ? config->tile_size : 32;
    if (c->tile_size < 8 || c->tile_size > 1024 || (c->tile_size & (c->tile_size - 1)) != 0) { delete c; return PT_ERR_INVALID_ARG; }  // power of two
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return PT_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = (uint32_t)prop.multiProcessorCount;
    if (config->stream || (config->flags & PT_FLAG_DEFAULT_STREAM)) {
        c->stream = reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(config->stream));  // 0 = legacy default stream
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PT_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess
        ) { pt_destroy(c); return PT_ERR_HIP; }
    c->n_lanes = config->frames_in_flight > 1 ? std::min(config->frames_in_flight, kMaxLanes) : ((config->flags & PT_FLAG_TWO_FRAMES_IN_FLIGHT) ? 2u : 1u);
    for (uint32_t i = 0; i < c->n_lanes; i++)
        if (hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming) != hipSuccess) { pt_destroy(c); return PT_ERR_HIP; }
    for (uint32_t i = 0; i < c->n_lanes; i++) {
        Lane& L = c->lanes[i];
        if (c->n_lanes == 1) L.stream = c->stream;
        else {
            // Up to three lanes get streams of the HIGHEST priority.  The runtime keeps one pool of hardware queues per priority
            // level and hands a new stream the least-used queue of its pool, so a lane created at the default priority shares the
            // 4 queues of that pool with every other stream of the process -- and a lane that lands on the hardware queue of the
            // CALLER's stream sits behind the completion waits of the frames before it (in-order queue): the same C2 frame took
            // 0.082 or 0.119 ms depending on how many streams the process had created before the context
            // (tools/experiments/qmap.py: 0 / 1 / 2 / 6 earlier streams fast, 3 / 4 / 5 / 7 / 8 slow).  In their own pool the lanes'
            // queues depend on nothing but their own number: 0.082 ms in all nine cases.  That pool serves three lanes well and
            // no more (4 / 6 / 8 lanes there: 256x256 frames 0.054 instead of 0.016-0.019 ms), so more than three lanes stay in
            // the default pool, where 6 are the best choice for small frames in a process that has few other streams.
            // PT_LANE_PRIORITY=0 creates every lane at the default priority (A/B runs).
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = the numerically lowest value = the highest priority
            const bool own_pool = knob_or(c->knobs.lane_priority, 1u) != 0 && hi < lo && c->n_lanes <= 3;
            const hipError_t e = own_pool ? hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
            if (e != hipSuccess) { pt_destroy(c); return PT_ERR_HIP; }
        }
        bool ok = true;
        for (auto& e : L.ev_poll) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (!ok || hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming) != hipSuccess
            || hipMalloc(&L.d_totals, 8 * sizeof(unsigned long long)) != hipSuccess
            || hipMalloc(&L.d_seg_counts, kMaxSegs * sizeof(uint32_t)) != hipSuccess
            || hipMemsetAsync(L.d_totals, 0, 8 * sizeof(unsigned long long), L.stream) != hipSuccess) { pt_destroy(c); return PT_ERR_HIP; }
    }
    *out_ctx = c;
    return PT_OK;
}

void pt_destroy(PtContext* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    if (c->comm && rccl().handle) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    free_textures(c);
    for (auto& L : c->lanes) {
        free_lane_buffers(L);
        free_lane_scene(L);
        free_dev(L.d_rot);
        if (L.h_rot_stage) (void)hipHostFree(L.h_rot_stage);
        if (L.ev_rot) (void)hipEventDestroy(L.ev_rot);
        if (L.ev_upload) (void)hipEventDestroy(L.ev_upload);
        for (auto& e : L.ev_poll) if (e) (void)hipEventDestroy(e);
        free_dev(L.d_counts); free_dev(L.d_totals); free_dev(L.d_seg_counts);
        if (L.h_counts) (void)hipHostFree(L.h_counts);
        if (L.h_prev_counts) (void)hipHostFree(L.h_prev_counts);
        if (L.ev_done) (void)hipEventDestroy(L.ev_done);
        if (L.stream && L.stream != c->stream) (void)hipStreamDestroy(L.stream);
    }
    free_dev(c->d_sph); free_dev(c->d_mats); free_dev(c->d_nodes); free_dev(c->d_wide); free_dev(c->d_sph_sorted); free_dev(c->d_sorted_id); free_dev(c->d_lights); free_dev(c->d_alpha_class); free_dev(c->d_leaf_ids);
    for (auto& b : c->beam.buf) {
        free_dev(b.d_lists);
        if (b.ev_ready) (void)hipEventDestroy(b.ev_ready);
    }
    if (c->beam.ev_last_use) (void)hipEventDestroy(c->beam.ev_last_use);
    if (c->beam.stream) (void)hipStreamDestroy(c->beam.stream);
    free_dev(c->d_out);
    for (auto& e : c->ev_in) if (e) (void)hipEventDestroy(e);
    if (c->gpu_builder) lbvh_gpu_destroy(c->gpu_builder);
    for (auto& p : c->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* pt_last_error(PtContext* c) { return c ? c->err.c_str() : "null context"; }

PtStatus pt_set_scene(PtContext* c, const PtSphere* spheres, const PtMaterial* materials, uint32_t n, const PtSceneData* sd)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!sd || (n != 0 && (!spheres || !materials))) return fail(c, PT_ERR_INVALID_ARG, "pt_set_scene: null pointer");
    if (n > (1u << 30)) return fail(c, PT_ERR_INVALID_ARG, "pt_set_scene: too many spheres");
    // An EMPTY scene is legal (a TLAS without instances: every ray misses, every pixel is the environment).  Inside, one sphere with a
    // NaN centre stands in: intersect_sphere's comparisons are all false for it, so no ray can hit it whatever its tmax, and the
    // single-sphere path of closest_hit has no tree that could look at its bounds.
    const bool empty = n == 0;
    const PtSphere nan_sphere{ std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::quiet_NaN(), 1.0f };
    PtMaterial blank_material{};
    blank_material.IOR = 1.0f;
    if (empty) { spheres = &nan_sphere; materials = &blank_material; n = 1; }
    for (uint32_t i = 0; i < n && !empty; i++)
        if (!(spheres[i].r > 0.0f) || !std::isfinite(spheres[i].r) || !std::isfinite(spheres[i].cx) || !std::isfinite(spheres[i].cy) || !std::isfinite(spheres[i].cz))
            return fail(c, PT_ERR_INVALID_ARG, "pt_set_scene: sphere " + std::to_string(i) + " has a non-finite centre or non-positive radius");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, sync_all(c));  // frames in flight still read the old scene
    float min_radius = std::numeric_limits<float>::infinity();  // (primary-beam lists of a moving camera: their slack is bounded by it)
    for (uint32_t i = 0; i < n && !empty; i++) min_radius = std::min(min_radius, spheres[i].r);
    // the emitters, in id order (LightPreparation::CountLights, Source/LightPreparation.ixx:52-70: objects with any emission > 0)
    std::vector<uint32_t> lights;
    for (uint32_t i = 0; i < n; i++) {
        const PtMaterial& m = materials[i];
        if (m.EmissiveStrength * m.EmissiveColor[0] > 0.0f || m.EmissiveStrength * m.EmissiveColor[1] > 0.0f || m.EmissiveStrength * m.EmissiveColor[2] > 0.0f)
            lights.push_back(i);
    }
    // Every allocation that can fail comes first, into temporaries: a call that runs out of memory leaves the previous scene intact
    // (nothing of the context has been touched yet).
    float4* new_sph = nullptr;
    float4* new_mats = nullptr;
    uint32_t* new_lights = nullptr;
    if (n != c->n || !c->d_sph || !c->d_mats) {
        if (hipMalloc(&new_sph, (size_t)n * sizeof(float4)) != hipSuccess || hipMalloc(&new_mats, (size_t)n * sizeof(PtMaterial)) != hipSuccess) {
            (void)hipGetLastError();
            free_dev(new_sph); free_dev(new_mats);
            return fail(c, PT_ERR_OOM, "pt_set_scene: out of device memory (the previous scene is unchanged)");
        }
    }
    if (!lights.empty() && hipMalloc(&new_lights, lights.size() * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        free_dev(new_sph); free_dev(new_mats);
        return fail(c, PT_ERR_OOM, "pt_set_scene: out of device memory (the previous scene is unchanged)");
    }
    c->empty_scene = empty;  // (after every check that can reject the call)
    c->min_radius = min_radius;
    if (new_sph) { free_dev(c->d_sph); free_dev(c->d_mats); c->d_sph = new_sph; c->d_mats = new_mats; }
    free_dev(c->d_lights);
    c->d_lights = new_lights;
    c->n_lights = (uint32_t)lights.size();
    c->n = n;
    c->h_sph.assign(spheres, spheres + n);
    PT_HIP(c, hipMemcpyAsync(c->d_sph, spheres, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    // device copy of the materials: the two padding words carry per-material constants of BSDFSample::Initialize
    // (dielectric F0 and 1/IOR), computed here once with the arithmetic the kernels would otherwise repeat per hit
    std::vector<PtMaterial> mats(materials, materials + n);
    for (auto& m : mats) {
        const float f0d = pt::dielectric_f0(m.IOR), inv_ior = 1.0f / m.IOR;
        std::memcpy(&m._pad[0], &f0d, 4);
        std::memcpy(&m._pad[1], &inv_ior, 4);
        m.AlphaMode &= ~kMaterialHasMaps;  // (bit 31 of the device copy is pt_set_textures' "has texture maps" flag; the alpha class comes from the host)
    }
    PT_HIP(c, hipMemcpyAsync(c->d_mats, mats.data(), (size_t)n * sizeof(PtMaterial), hipMemcpyHostToDevice, c->stream));
    if (c->n_lights) PT_HIP(c, hipMemcpyAsync(c->d_lights, lights.data(), lights.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));  // caller-owned host memory may be released on return
    c->sd = *sd;
    c->scene_set = true;
    c->accel_valid = false;
    c->scene_gen++;
    free_textures(c);  // texture maps are per sphere: a new scene starts untextured
    // alpha-tested hits: the spheres that are not Opaque (Scene.ixx:242-243), classified now from their constant alpha and again when
    // texture maps arrive (pt_set_textures)
    c->alpha_mats.clear();
    for (uint32_t i = 0; i < n && !empty; i++)
        if (materials[i].AlphaMode != PT_ALPHA_OPAQUE) {
            PtContext::AlphaMat am{};
            am.id = i; am.cutoff = materials[i].AlphaCutoff; am.base_map = ~0u;
            for (int k = 0; k < 4; k++) am.base[k] = materials[i].BaseColor[k];
            c->alpha_mats.push_back(am);
        }
    if (PtStatus st = update_alpha_classes(c); st != PT_OK) return st;
    for (auto& L : c->lanes) { L.scene_private = false; L.upload_pending = false; L.needs_refit = false; L.sph_gen = 0; }  // every lane renders the new master scene
    c->sph_gen = 0;
    c->latest_lane = -1;
    return PT_OK;
}

PtStatus pt_build_accel(PtContext* c, PtAccelInfo* info)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->scene_set) return fail(c, PT_ERR_STATE, "pt_build_accel: no scene");
    const RoctxRange range(c, "pt_build_accel");
    PT_HIP(c, hipSetDevice(c->device));
    const uint32_t n = c->n;
    PT_HIP(c, sync_all(c));
    if (c->n_nodes != (n > 1 ? n - 1 : 0) || !c->d_sph_sorted) {
        free_dev(c->d_nodes); free_dev(c->d_sph_sorted); free_dev(c->d_sorted_id);
        PT_HIP(c, hipMalloc(&c->d_nodes, (size_t)std::max(1u, n - 1) * sizeof(PtBvhNode)));
        PT_HIP(c, hipMalloc(&c->d_sph_sorted, (size_t)n * sizeof(float4)));
        PT_HIP(c, hipMalloc(&c->d_sorted_id, (size_t)n * sizeof(uint32_t)));
    }
    c->n_nodes = n > 1 ? n - 1 : 0;
    float build_ms = 0;
    uint32_t builder_kind;
    if ((c->flags & PT_FLAG_HOST_LBVH) || !lbvh_gpu_available()) {
        auto t0 = std::chrono::steady_clock::now();
        build_lbvh_host(c->h_sph.data(), n, c->lbvh);
        auto t1 = std::chrono::steady_clock::now();
        build_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
        if (c->n_nodes) PT_HIP(c, hipMemcpyAsync(c->d_nodes, c->lbvh.nodes.data(), (size_t)c->n_nodes * sizeof(PtBvhNode), hipMemcpyHostToDevice, c->stream));
        PT_HIP(c, hipMemcpyAsync(c->d_sph_sorted, c->lbvh.sorted.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        PT_HIP(c, hipMemcpyAsync(c->d_sorted_id, c->lbvh.sorted_id.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
        c->depth = c->lbvh.depth;
        builder_kind = PT_BUILDER_HOST_LBVH;
    } else {
        if (!c->gpu_builder) {
            c->gpu_builder = lbvh_gpu_create();
            if (!c->gpu_builder) return fail(c, PT_ERR_OOM, "pt_build_accel: cannot create the device LBVH builder");
        }
        LbvhGpuInfo gi{};
        hipError_t e;
        // Small scenes get a SAH topology from the host (the PREFER_FAST_TRACE analogue: 7 % fewer cycles per C2 frame than the
        // Morton-order tree; the build is tens of microseconds at this size); larger ones the device LBVH, whose cost stays
        // O(n) on the GPU.  PT_FLAG_FAST_BUILD / PT_SAH=0 keep the LBVH everywhere.
        const bool sah = !(c->flags & PT_FLAG_FAST_BUILD) && n > 2 && n <= knob_or(c->knobs.sah_max_spheres, kSahMaxSpheres) && knob_or(c->knobs.sah, 1u) != 0;
        if (sah) {
            auto t0 = std::chrono::steady_clock::now();
            build_sah_host(c->h_sph.data(), n, c->lbvh);
            auto t1 = std::chrono::steady_clock::now();
            e = lbvh_gpu_adopt(c->gpu_builder, c->d_sph, n, c->lbvh.nodes.data(), c->lbvh.sorted_id.data(), c->lbvh.depth,
                               reinterpret_cast<PtBvhNode*>(c->d_nodes), c->d_sph_sorted, c->d_sorted_id, c->stream, &gi);
            gi.build_ms += std::chrono::duration<float, std::milli>(t1 - t0).count();
            builder_kind = PT_BUILDER_HOST_SAH;
        } else {
            e = lbvh_gpu_build(c->gpu_builder, c->d_sph, n, reinterpret_cast<PtBvhNode*>(c->d_nodes), c->d_sph_sorted, c->d_sorted_id, c->stream, &gi);
            builder_kind = PT_BUILDER_DEVICE_LBVH;
        }
        if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP, std::string("device BVH build: ") + hipGetErrorString(e));
        c->depth = gi.depth;
        build_ms = gi.build_ms;
        c->lbvh = LbvhResult{};
        c->lbvh.depth = gi.depth;
        c->lbvh.pad = gi.pad;
        for (int a = 0; a < 3; a++) { c->lbvh.bounds_min[a] = gi.bounds_min[a]; c->lbvh.bounds_max[a] = gi.bounds_max[a]; }
    }
    // stage the BVH in LDS when scene + stacks leave room for two workgroups per CU
    const uint32_t scene_bytes = traverse_lds_bytes_for(c->n_nodes, n, 0, true);
    c->lds_scene = !(c->flags & PT_FLAG_NO_LDS_SCENE) && scene_bytes <= kLdsSceneBudget
                   && traverse_lds_bytes_for(c->n_nodes, n, std::max(1u, c->depth), true) <= kMaxLdsBytes / 2;
    // Scenes that traverse global memory get the 4-wide view of the tree: half the dependent node fetches per ray (the bound of
    // the 2^20-sphere scene).  PT_WIDE=0 keeps the binary walk, for A/B runs.
    free_dev(c->d_wide);
    if (!c->lds_scene && c->n_nodes > 1 && knob_or(c->knobs.wide, 1u) != 0) {
        PT_HIP(c, hipMalloc(&c->d_wide, (size_t)c->n_nodes * 4u * sizeof(float4)));
        PT_HIP(c, lbvh_gpu_collapse4(reinterpret_cast<const PtBvhNode*>(c->d_nodes), c->n_nodes, c->d_wide, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->accel_valid = true;
    c->scene_gen++;
    if (PtStatus st = refresh_leaf_ids(c); st != PT_OK) return st;
    if (info) {
        std::memset(info, 0, sizeof *info);
        info->leaf_count = c->empty_scene ? 0u : n;
        info->node_count = c->n_nodes;
        info->depth = c->depth;
        info->lds_resident = c->lds_scene ? 1u : 0u;
        for (int a = 0; a < 3; a++) { info->bounds_min[a] = c->empty_scene ? 0.0f : c->lbvh.bounds_min[a]; info->bounds_max[a] = c->empty_scene ? 0.0f : c->lbvh.bounds_max[a]; }
        info->build_ms = build_ms;
        info->builder = builder_kind;
    }
    return PT_OK;
}

PtStatus pt_update_spheres(PtContext* c, const PtSphere* spheres, uint32_t n)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->empty_scene && n == 0 && c->scene_set) return PT_OK;  // nothing to move
    if (c->empty_scene) return fail(c, PT_ERR_INVALID_ARG, "pt_update_spheres: the sphere count must not change (the scene is empty; use pt_set_scene)");
    if (!spheres) return fail(c, PT_ERR_INVALID_ARG, "pt_update_spheres: null pointer");
    if (!c->scene_set || !c->accel_valid) return fail(c, PT_ERR_STATE, "pt_update_spheres: pt_set_scene + pt_build_accel first");
    if (n != c->n) return fail(c, PT_ERR_INVALID_ARG, "pt_update_spheres: the sphere count must not change (use pt_set_scene)");
    if (!c->gpu_builder) return fail(c, PT_ERR_UNSUPPORTED, "pt_update_spheres needs the device LBVH builder (no PT_FLAG_HOST_LBVH)");
    for (uint32_t i = 0; i < n; i++)
        if (!(spheres[i].r > 0.0f) || !std::isfinite(spheres[i].r) || !std::isfinite(spheres[i].cx) || !std::isfinite(spheres[i].cy) || !std::isfinite(spheres[i].cz))
            return fail(c, PT_ERR_INVALID_ARG, "pt_update_spheres: sphere " + std::to_string(i) + " has a non-finite centre or non-positive radius");
    PT_HIP(c, hipSetDevice(c->device));
    Lane& L = c->lanes[c->next_lane];  // the lane the next render call will use
    c->scene_gen++;
    if (PtStatus st = stage_spheres_on_lane(c, L, spheres); st != PT_OK) return st;
    // the other lanes take these spheres over from this lane's staging buffer when they render next (sync_lane_spheres)
    L.sph_gen = ++c->sph_gen;
    c->latest_lane = (int)c->next_lane;
    return PT_OK;
}


PtStatus pt_refit_accel(PtContext* c)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->empty_scene && c->accel_valid) return PT_OK;  // no boxes
    Lane& L = c->lanes[c->next_lane];
    if (!c->accel_valid || !L.scene_private) return fail(c, PT_ERR_STATE, "pt_refit_accel: call pt_update_spheres first");
    const RoctxRange range(c, "pt_refit_accel");
    PT_HIP(c, hipSetDevice(c->device));
    return refit_lane(c, L);
}

PtStatus pt_set_camera(PtContext* c, const PtCamera* camera)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!camera) return fail(c, PT_ERR_INVALID_ARG, "pt_set_camera: null camera");
    c->cam = *camera;
    c->cam_set = true;
    return PT_OK;
}

PtStatus pt_set_constants(PtContext* c, const PtGraphicsSettings* gs)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!gs) return fail(c, PT_ERR_INVALID_ARG, "pt_set_constants: null settings");
    if (gs->RenderSize[0] == 0 || gs->RenderSize[1] == 0 || gs->RenderSize[0] > 65535u || gs->RenderSize[1] > 65535u)
        return fail(c, PT_ERR_INVALID_ARG, "pt_set_constants: RenderSize must be in [1, 65535]^2 (the RNG seed packs (x << 16) | y)");
    if (gs->SamplesPerPixel == 0 || gs->SamplesPerPixel > 65535u) return fail(c, PT_ERR_INVALID_ARG, "pt_set_constants: SamplesPerPixel must be in [1, 65535]");
    if (gs->Bounces > 250u) return fail(c, PT_ERR_INVALID_ARG, "pt_set_constants: Bounces must be <= 250");
    if (gs->Denoiser) return fail(c, PT_ERR_UNSUPPORTED, "Denoiser must be 0 == Denoiser::None");
    c->gs = *gs;
    c->gs_set = true;
    return PT_OK;
}

PtStatus pt_set_partition(PtContext* c, uint32_t rank, uint32_t world)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (world == 0 || rank >= world) return fail(c, PT_ERR_INVALID_ARG, "pt_set_partition: need rank < world");
    c->rank = rank;
    c->world = world;
    c->part_first = rank; c->part_run = 1; c->part_stride = world;
    return PT_OK;
}

PtStatus pt_set_partition_ex(PtContext* c, uint32_t first, uint32_t run, uint32_t stride)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (stride == 0 || (uint64_t)first + run > stride) return fail(c, PT_ERR_INVALID_ARG, "pt_set_partition_ex: need first + run <= stride, stride > 0");
    c->part_first = first; c->part_run = run; c->part_stride = stride;
    c->rank = 0; c->world = 1;  // pt_tiles_count(rank) / pt_unpack_tiles describe the plain interleave only
    return PT_OK;
}

uint32_t pt_tiles_count(PtContext* c, uint32_t rank)
{
    if (!c || !c->gs_set || rank >= c->world) return 0;
    return count_tiles(frame_tiles(c), rank, 1, c->world);
}

uint32_t pt_tiles_count_ex(PtContext* c, uint32_t first, uint32_t run, uint32_t stride)
{
    if (!c || !c->gs_set || stride == 0 || (uint64_t)first + run > stride) return 0;
    return count_tiles(frame_tiles(c), first, run, stride);
}

PtStatus pt_render(PtContext* c, const PtRect* rect, void* out, int out_is_device, PtStats* stats)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!out) return fail(c, PT_ERR_INVALID_ARG, "pt_render: null output");
    PtStatus st = validate_frame(c);
    if (st != PT_OK) return st;
    PT_HIP(c, hipSetDevice(c->device));
    PtRect r = rect ? *rect : PtRect{ 0, 0, c->gs.RenderSize[0], c->gs.RenderSize[1] };
    if (r.w == 0 || r.h == 0 || r.x >= c->gs.RenderSize[0] || r.w > c->gs.RenderSize[0] - r.x || r.y >= c->gs.RenderSize[1] || r.h > c->gs.RenderSize[1] - r.y)
        return fail(c, PT_ERR_INVALID_ARG, "pt_render: rect is empty or outside RenderSize");
    PixelMap pm{};
    pm.mode = 0;
    pm.img_w = c->gs.RenderSize[0]; pm.img_h = c->gs.RenderSize[1];
    pm.rx = r.x; pm.ry = r.y; pm.rw = r.w; pm.rh = r.h;
    pm.blocks_x = (r.w + 7) / 8;
    pm.inv_blocks_x = 1.0f / (float)pm.blocks_x;
    const uint64_t slots = (uint64_t)pm.blocks_x * ((r.h + 7) / 8) * 64ull;
    pm.exact_div = (slots >> 6) >= (1ull << 22) ? 1u : 0u;  // quotient estimate error stays below one
    if (slots > 0xFFFFFFFFull) return fail(c, PT_ERR_INVALID_ARG, "pt_render: rect too large");
    pm.n_slots = (uint32_t)slots;
    float4* dev_out = static_cast<float4*>(out);
    const size_t out_px = (size_t)r.w * r.h;
    if (!out_is_device) {
        if (out_px > c->cap_out) {
            PT_HIP(c, sync_all(c));
            free_dev(c->d_out);
            PT_HIP(c, hipMalloc(&c->d_out, out_px * sizeof(float4)));
            c->cap_out = out_px;
        }
        dev_out = c->d_out;
    }
    st = render_common(c, pm, (uint64_t)r.w * r.h, dev_out, stats);
    if (st != PT_OK) return st;
    if (!out_is_device) {
        PT_HIP(c, hipMemcpyAsync(out, dev_out, out_px * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return PT_OK;
}

PtStatus pt_render_tiles(PtContext* c, void* out_device_packed, PtStats* stats)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!out_device_packed) return fail(c, PT_ERR_INVALID_ARG, "pt_render_tiles: null output");
    PtStatus st = validate_frame(c);
    if (st != PT_OK) return st;
    PT_HIP(c, hipSetDevice(c->device));
    const uint32_t ts = c->tile_size, w = c->gs.RenderSize[0], h = c->gs.RenderSize[1];
    PixelMap pm{};
    pm.mode = 1;
    pm.img_w = w; pm.img_h = h;
    pm.ts = ts;
    pm.ts_shift = (uint32_t)__builtin_ctz(ts);
    pm.tiles_x = (w + ts - 1) / ts;
    pm.inv_tiles_x = 1.0f / (float)pm.tiles_x;
    pm.tiles_total = pm.tiles_x * ((h + ts - 1) / ts);
    pm.first = c->part_first; pm.run = c->part_run; pm.stride = c->part_stride;
    pm.inv_run = pm.run ? 1.0f / (float)pm.run : 0.0f;
    const uint64_t slots = (uint64_t)count_tiles(pm.tiles_total, pm.first, pm.run, pm.stride) * ts * ts;
    if (slots == 0) {  // this rank owns no tile of the frame
        if (stats) std::memset(stats, 0, sizeof *stats);
        return PT_OK;
    }
    if (slots > 0xFFFFFFFFull) return fail(c, PT_ERR_INVALID_ARG, "pt_render_tiles: too many pixels");
    pm.n_slots = (uint32_t)slots;
    pm.exact_div = (slots >> (2u * pm.ts_shift)) >= (1ull << 22) ? 1u : 0u;
    return render_common(c, pm, count_tile_pixels(w, h, ts, pm.first, pm.run, pm.stride), static_cast<float4*>(out_device_packed), stats);
}

PtStatus pt_unpack_tiles(PtContext* c, const void* gathered, uint32_t max_tiles_per_rank, void* frame)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!gathered || !frame || !c->gs_set) return fail(c, PT_ERR_INVALID_ARG, "pt_unpack_tiles: null pointer or no constants");
    if (max_tiles_per_rank < pt_tiles_count(c, 0)) return fail(c, PT_ERR_INVALID_ARG, "pt_unpack_tiles: max_tiles_per_rank too small");
    PT_HIP(c, hipSetDevice(c->device));
    const uint32_t ts = c->tile_size, w = c->gs.RenderSize[0], h = c->gs.RenderSize[1];
    PT_HIP(c, launch_unpack_tiles(static_cast<const float4*>(gathered), static_cast<float4*>(frame), w, h, ts, (w + ts - 1) / ts, 0, 1, c->world,
                                  c->world, (uint64_t)max_tiles_per_rank * ts * ts, false, c->stream));
    return PT_OK;
}

static PtStatus unpack_ex(PtContext* c, const void* packed, uint64_t part_stride_px, uint32_t n_parts, uint32_t first0, uint32_t run,
                          uint32_t stride, void* frame, bool rgb);

PtStatus pt_unpack_tiles_ex(PtContext* c, const void* packed, uint64_t part_stride_px, uint32_t n_parts, uint32_t first0, uint32_t run,
                            uint32_t stride, void* frame)
{
    return unpack_ex(c, packed, part_stride_px, n_parts, first0, run, stride, frame, false);
}

PtStatus pt_unpack_tiles_rgb(PtContext* c, const void* packed, uint64_t part_stride_px, uint32_t n_parts, uint32_t first0, uint32_t run,
                             uint32_t stride, void* frame)
{
    return unpack_ex(c, packed, part_stride_px, n_parts, first0, run, stride, frame, true);
}

PtStatus pt_pack_rgb(PtContext* c, const void* src, uint64_t n_pixels, void* dst)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!src || !dst) return fail(c, PT_ERR_INVALID_ARG, "pt_pack_rgb: null pointer");
    if (n_pixels == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, launch_pack_rgb(static_cast<const float4*>(src), static_cast<float*>(dst), n_pixels, c->stream));
    return PT_OK;
}

static PtStatus unpack_ex(PtContext* c, const void* packed, uint64_t part_stride_px, uint32_t n_parts, uint32_t first0, uint32_t run,
                          uint32_t stride, void* frame, bool rgb)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!packed || !frame || !c->gs_set) return fail(c, PT_ERR_INVALID_ARG, "pt_unpack_tiles_ex: null pointer or no constants");
    if (stride == 0 || run == 0 || n_parts == 0 || (uint64_t)first0 + (uint64_t)n_parts * run > stride)
        return fail(c, PT_ERR_INVALID_ARG, "pt_unpack_tiles_ex: need run, n_parts > 0 and first0 + n_parts * run <= stride");
    const uint32_t ts = c->tile_size, w = c->gs.RenderSize[0], h = c->gs.RenderSize[1];
    // the first part owns at least as many tiles as any later one
    if (n_parts > 1 && part_stride_px < (uint64_t)count_tiles(frame_tiles(c), first0, run, stride) * ts * ts)
        return fail(c, PT_ERR_INVALID_ARG, "pt_unpack_tiles_ex: part_stride_px smaller than one part's tiles");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, launch_unpack_tiles(static_cast<const float4*>(packed), static_cast<float4*>(frame), w, h, ts, (w + ts - 1) / ts, first0, run, stride,
                                  n_parts, part_stride_px, rgb, c->stream));
    return PT_OK;
}

PtStatus pt_set_textures(PtContext* c, const PtTexture* textures, uint32_t n_textures, const PtObjectTextures* object_textures, const float* rotations)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->scene_set) return fail(c, PT_ERR_STATE, "pt_set_textures: call pt_set_scene first");
    PT_HIP(c, hipSetDevice(c->device));
    if (n_textures == 0) {  // drops the table
        PT_HIP(c, sync_all(c));
        free_textures(c);
        for (auto& am : c->alpha_mats) am.base_map = ~0u;
        PT_HIP(c, launch_material_map_flags(c->d_mats, nullptr, c->n, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
        return update_alpha_classes(c);
    }
    if (!textures) return fail(c, PT_ERR_INVALID_ARG, "pt_set_textures: null pointer");
    const uint32_t n = c->n;
    // validate before touching the device or the table in use: a rejected call leaves the previous textures in place
    for (uint32_t t = 0; t < n_textures; t++) {
        const PtTexture& tx = textures[t];
        if (!tx.Pixels || tx.Width == 0 || tx.Height == 0 || tx.Width > 16384 || tx.Height > 16384 || tx.Format > PT_TEXTURE_RGBA32_FLOAT)
            return fail(c, PT_ERR_INVALID_ARG, "pt_set_textures: bad texture (null pixels, size outside 1..16384, or unknown format)");
    }
    std::vector<uint32_t> maps((size_t)n * 8u);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t any = 0;
        for (uint32_t k = 0; k < PT_TEXTURE_MAP_COUNT; k++) {
            const PtTextureMapInfo none{ ~0u, 0u, { 0u, 0u } };
            const PtTextureMapInfo& mi = (object_textures && !c->empty_scene) ? object_textures[i].Maps[k] : none;
            if (mi.Descriptor != ~0u) {
                if (mi.Descriptor >= n_textures) return fail(c, PT_ERR_INVALID_ARG, "pt_set_textures: Descriptor out of range");
                if (mi.TextureCoordinateIndex != 0) return fail(c, PT_ERR_UNSUPPORTED, "pt_set_textures: spheres have one texture-coordinate set (index 0)");
                any = 1;
            }
            maps[(size_t)i * 8u + k] = mi.Descriptor;
        }
        maps[(size_t)i * 8u + 7u] = any;
    }
    PT_HIP(c, sync_all(c));  // frames in flight may still sample the old table
    free_textures(c);
    // 8-bit texels -> linear float4 (the conversion D3D's sampler does per fetch, done once; sRGB through from_srgb)
    float unorm_lut[256], srgb_lut[256];
    for (int v = 0; v < 256; v++) { unorm_lut[v] = (float)v * (1.0f / 255.0f); srgb_lut[v] = pt::from_srgb(unorm_lut[v]); }
    std::vector<TexView> views(n_textures);
    std::vector<float4> texels;
    c->d_tex_images.assign(n_textures, nullptr);
    for (uint32_t t = 0; t < n_textures; t++) {
        const PtTexture& tx = textures[t];
        const size_t count = (size_t)tx.Width * tx.Height;
        const void* src = tx.Pixels;  // RGBA32_FLOAT is the device layout already
        if (tx.Format != PT_TEXTURE_RGBA32_FLOAT) {
            texels.resize(count);
            const uint8_t* px = static_cast<const uint8_t*>(tx.Pixels);
            const float* lut = tx.Format == PT_TEXTURE_RGBA8_UNORM_SRGB ? srgb_lut : unorm_lut;
            for (size_t i = 0; i < count; i++)
                texels[i] = make_float4(lut[px[4 * i]], lut[px[4 * i + 1]], lut[px[4 * i + 2]], unorm_lut[px[4 * i + 3]]);
            src = texels.data();
        }
        PT_HIP(c, hipMalloc(&c->d_tex_images[t], count * sizeof(float4)));
        PT_HIP(c, hipMemcpy(c->d_tex_images[t], src, count * sizeof(float4), hipMemcpyHostToDevice));
        views[t].texels = c->d_tex_images[t]; views[t].w = tx.Width; views[t].h = tx.Height;
        c->tex_dims.emplace_back(tx.Width, tx.Height);
    }
    PT_HIP(c, hipMalloc(&c->d_tex, n_textures * sizeof(TexView)));
    PT_HIP(c, hipMemcpy(c->d_tex, views.data(), n_textures * sizeof(TexView), hipMemcpyHostToDevice));
    PT_HIP(c, hipMalloc(&c->d_tex_maps, maps.size() * sizeof(uint32_t)));
    PT_HIP(c, hipMemcpy(c->d_tex_maps, maps.data(), maps.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    PT_HIP(c, hipMalloc(&c->d_rot, (size_t)n * sizeof(float4)));
    c->h_rot.assign((size_t)n, make_float4(0.f, 0.f, 0.f, 1.f));
    if (rotations)
        for (uint32_t i = 0; i < n && !c->empty_scene; i++) c->h_rot[i] = make_float4(rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]);
    PT_HIP(c, hipMemcpy(c->d_rot, c->h_rot.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice));  // (everything was synchronised above)
    c->rot_master_gen = ++c->rot_gen;  // the master copy is current; lanes take private copies from the next pt_update_rotations on
    c->has_textures = true;
    PT_HIP(c, launch_material_map_flags(c->d_mats, c->d_tex_maps, n, c->stream));  // kMaterialHasMaps in the device materials
    PT_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& am : c->alpha_mats) am.base_map = maps[(size_t)am.id * 8u + kMapBaseColor];
    return update_alpha_classes(c);  // a base-colour map turns a non-opaque sphere's alpha test into a per-crossing one
}

PtStatus pt_update_rotations(PtContext* c, const float* rotations, uint32_t n)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->has_textures) return fail(c, PT_ERR_STATE, "pt_update_rotations: the scene has no textures (rotations only orient texture coordinates)");
    if (c->empty_scene) return n == 0 ? PT_OK : fail(c, PT_ERR_INVALID_ARG, "pt_update_rotations: count differs from the scene's sphere count");
    if (n != c->n) return fail(c, PT_ERR_INVALID_ARG, "pt_update_rotations: count differs from the scene's sphere count");
    // No device work and no wait here: the frames in flight keep the rotations they were submitted with; every later render
    // call uploads these into its lane's own copy on its own stream (sync_lane_rotations), like pt_update_spheres does.
    for (uint32_t i = 0; i < n; i++)
        c->h_rot[i] = rotations ? make_float4(rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]) : make_float4(0.f, 0.f, 0.f, 1.f);
    c->rot_gen++;
    return PT_OK;
}

PtStatus pt_tonemap(PtContext* c, const void* hdr, uint32_t n_pixels, const PtToneMapParams* params, void* out)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!hdr || !out || !params) return fail(c, PT_ERR_INVALID_ARG, "pt_tonemap: null pointer");
    if (params->Operator > kToneACESFilmic || params->TransferFunction > kTransferST2084 || params->ColorRotation > kRotate709toP3D65)
        return fail(c, PT_ERR_INVALID_ARG, "pt_tonemap: unknown operator / transfer function / colour rotation");
    if (n_pixels == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, launch_tonemap(static_cast<const float4*>(hdr), static_cast<uint32_t*>(out), n_pixels, *params, c->stream));
    return PT_OK;
}

PtStatus pt_accumulate(PtContext* c, void* accum, const void* radiance, uint32_t n_pixels, uint32_t frames_accumulated)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!accum || !radiance) return fail(c, PT_ERR_INVALID_ARG, "pt_accumulate: null pointer");
    if (frames_accumulated == 0xFFFFFFFFu) return fail(c, PT_ERR_INVALID_ARG, "pt_accumulate: frame count overflow");
    if (n_pixels == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, launch_accumulate(static_cast<float4*>(accum), static_cast<const float4*>(radiance), n_pixels, frames_accumulated, c->stream));
    return PT_OK;
}

static PtStatus trace_rays_impl(PtContext* c, const float* origins, const float* directions, uint32_t n, float tmin, int use_bvh, float* out_t,
                                uint32_t* out_id, uint32_t* out_visits)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!origins || !directions || !out_t || !out_id) return fail(c, PT_ERR_INVALID_ARG, "pt_trace_rays: null pointer");
    if (!c->scene_set || !c->accel_valid) return fail(c, PT_ERR_STATE, "pt_trace_rays: scene / accel not ready");
    if (n == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    uint32_t* d_id = nullptr;
    uint2* d_v = nullptr;
    PtStatus st = PT_OK;
    auto cleanup = [&] { free_dev(d_o); free_dev(d_d); free_dev(d_t); free_dev(d_id); free_dev(d_v); };
    hipError_t e;
    if ((e = hipMalloc(&d_o, (size_t)n * 12)) != hipSuccess || (e = hipMalloc(&d_d, (size_t)n * 12)) != hipSuccess
        || (e = hipMalloc(&d_t, (size_t)n * 4)) != hipSuccess || (e = hipMalloc(&d_id, (size_t)n * 4)) != hipSuccess
        || (out_visits && use_bvh && (e = hipMalloc(&d_v, (size_t)n * 8)) != hipSuccess)) {
        cleanup();
        return fail(c, PT_ERR_OOM, std::string("pt_trace_rays: ") + hipGetErrorString(e));
    }
    const SceneView sv = make_scene_view(c);
    if ((e = hipMemcpyAsync(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice, c->stream)) != hipSuccess
        || (e = hipMemcpyAsync(d_d, directions, (size_t)n * 12, hipMemcpyHostToDevice, c->stream)) != hipSuccess
        || (e = launch_trace(sv, d_o, d_d, n, tmin, use_bvh, d_t, d_id, d_v, c->stream)) != hipSuccess
        || (e = hipMemcpyAsync(out_t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess
        || (e = hipMemcpyAsync(out_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess
        || (d_v && (e = hipMemcpyAsync(out_visits, d_v, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        || (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
        st = fail(c, PT_ERR_HIP, std::string("pt_trace_rays: ") + hipGetErrorString(e));
    }
    cleanup();
    return st;
}

PtStatus pt_trace_rays(PtContext* c, const float* origins, const float* directions, uint32_t n, float tmin, int use_bvh, float* out_t, uint32_t* out_id)
{
    return trace_rays_impl(c, origins, directions, n, tmin, use_bvh, out_t, out_id, nullptr);
}

PtStatus pt_trace_rays_stats(PtContext* c, const float* origins, const float* directions, uint32_t n, float tmin, float* out_t, uint32_t* out_id,
                             uint32_t* out_visits)
{
    if (c && !out_visits) return fail(c, PT_ERR_INVALID_ARG, "pt_trace_rays_stats: null pointer");
    return trace_rays_impl(c, origins, directions, n, tmin, 1, out_t, out_id, out_visits);
}

PtStatus pt_accel_download(PtContext* c, PtBvhNode* nodes, uint32_t capacity)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->accel_valid) return fail(c, PT_ERR_STATE, "pt_accel_download: no accel");
    if (!nodes || capacity < c->n_nodes) return fail(c, PT_ERR_INVALID_ARG, "pt_accel_download: buffer too small");
    if (c->n_nodes == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemcpyAsync(nodes, c->d_nodes, (size_t)c->n_nodes * sizeof(PtBvhNode), hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

PtStatus pt_accel_download_order(PtContext* c, uint32_t* sorted_id, uint32_t capacity)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->accel_valid) return fail(c, PT_ERR_STATE, "pt_accel_download_order: no accel");
    if (!sorted_id || capacity < c->n) return fail(c, PT_ERR_INVALID_ARG, "pt_accel_download_order: buffer too small");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemcpyAsync(sorted_id, c->d_sorted_id, (size_t)c->n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

PtStatus pt_set_profiling(PtContext* c, int enabled)
{
    if (!c) return PT_ERR_INVALID_ARG;
    PT_HIP(c, sync_all(c));
    c->profiling = enabled != 0;
    c->ev_used = 0;
    return PT_OK;
}

PtStatus pt_get_profile(PtContext* c, PtStats* profile, int reset)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!profile) return fail(c, PT_ERR_INVALID_ARG, "pt_get_profile: null output");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, sync_all(c));
    std::memset(profile, 0, sizeof *profile);
    sum_events(c, 0, c->ev_used, profile);
    if (reset) c->ev_used = 0;
    return PT_OK;
}

PtStatus pt_get_totals(PtContext* c, PtStats* totals, int reset)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!totals) return fail(c, PT_ERR_INVALID_ARG, "pt_get_totals: null output");
    PT_HIP(c, hipSetDevice(c->device));
    unsigned long long secondary = 0, first_pass = 0, node_visits = 0, sphere_tests = 0;
    for (uint32_t i = 0; i < c->n_lanes; i++) {
        Lane& L = c->lanes[i];
        unsigned long long s[8] = {};
        PT_HIP(c, flush_all_counters(L));  // fold the frames still sitting in the per-frame counters
        PT_HIP(c, hipMemcpyAsync(s, L.d_totals, sizeof s, hipMemcpyDeviceToHost, L.stream));
        PT_HIP(c, hipStreamSynchronize(L.stream));
        secondary += s[0];
        first_pass += s[4];
        node_visits += s[6];
        sphere_tests += s[7];
        if (reset) {
            PT_HIP(c, hipMemsetAsync(L.d_totals, 0, 2 * sizeof(unsigned long long), L.stream));
            PT_HIP(c, hipMemsetAsync(L.d_totals + 4, 0, 4 * sizeof(unsigned long long), L.stream));
        }
    }
    std::memset(totals, 0, sizeof *totals);
    totals->rays_first_pass_inline = first_pass;
    totals->node_visits = node_visits;
    totals->sphere_tests = sphere_tests;
    totals->rays = c->tot_pixels + secondary;
    totals->paths = c->tot_paths;
    totals->pixels = c->tot_pixels;
    totals->bytes_algorithmic = c->tot_sec_coeff * secondary + c->tot_fixed_bytes;
    totals->beams_used = c->tot_beam_frames;
    if (reset) { c->tot_pixels = c->tot_paths = c->tot_fixed_bytes = 0; c->tot_beam_frames = 0; }
    return PT_OK;
}

PtStatus pt_get_queue_sizes(PtContext* c, uint32_t* sizes, uint32_t capacity, uint32_t* n)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!sizes || !n) return fail(c, PT_ERR_INVALID_ARG, "pt_get_queue_sizes: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, sync_all(c));
    *n = 0;
    Lane& L = c->lanes[c->last_lane];
    if (!L.h_prev_counts || !L.prev_signature) return PT_OK;
    // the latest frame's sizes are published when its counters are folded: do that now if no later frame has
    PT_HIP(c, flush_all_counters(L));
    PT_HIP(c, hipStreamSynchronize(L.stream));
    uint32_t k = 0;
    for (; k < L.cap_counts && k < capacity; k++) {
        sizes[k] = L.h_prev_counts[k];
        if (k > 0 && sizes[k] == 0) break;
    }
    *n = k;
    return PT_OK;
}

PtStatus pt_device_alloc(PtContext* c, uint64_t bytes, void** out_device)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!out_device || bytes == 0) return fail(c, PT_ERR_INVALID_ARG, "pt_device_alloc: null pointer or zero size");
    *out_device = nullptr;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMalloc(out_device, (size_t)bytes));
    return PT_OK;
}

PtStatus pt_device_free(PtContext* c, void* device)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!device) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, sync_all(c));
    PT_HIP(c, hipFree(device));
    return PT_OK;
}

PtStatus pt_download(PtContext* c, const void* device, void* host, uint64_t bytes)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!device || !host) return fail(c, PT_ERR_INVALID_ARG, "pt_download: null pointer");
    if (bytes == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemcpyAsync(host, device, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// ---- multi-GPU exchange (SURVEY 8b pt_gather / 8e) ---------------------------------------------------------------------
#define PT_NCCL(ctx, expr)                                                                                       \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess) return fail(ctx, PT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

PtStatus pt_comm_unique_id(void* id_out)
{
    if (!id_out) return PT_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must match ncclUniqueId");
    Rccl& R = rccl();
    if (!R.handle) return PT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (R.GetUniqueId(&id) != ncclSuccess) return PT_ERR_HIP;
    std::memcpy(id_out, &id, sizeof id);
    return PT_OK;
}

PtStatus pt_comm_init(PtContext* c, const void* id, uint32_t rank, uint32_t world)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!id || world == 0 || rank >= world) return fail(c, PT_ERR_INVALID_ARG, "pt_comm_init: null id or rank >= world");
    if (c->comm) return fail(c, PT_ERR_STATE, "pt_comm_init: this context already has a communicator (pt_comm_destroy first)");
    Rccl& R = rccl();
    if (!R.handle) return fail(c, PT_ERR_UNSUPPORTED, "pt_comm_init: " + R.error);
    PT_HIP(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    PT_NCCL(c, R.CommInitRank(&c->comm, (int)world, uid, (int)rank));
    c->comm_rank = rank;
    c->comm_world = world;
    return PT_OK;
}

PtStatus pt_comm_destroy(PtContext* c)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->comm) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, sync_all(c));
    PT_NCCL(c, rccl().CommDestroy(c->comm));
    c->comm = nullptr;
    c->comm_rank = 0; c->comm_world = 1;
    return PT_OK;
}

PtStatus pt_gather(PtContext* c, const void* send_device, void* recv_device, uint64_t bytes, uint32_t root)
{
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->comm) return fail(c, PT_ERR_STATE, "pt_gather: no communicator (pt_comm_init)");
    if (root >= c->comm_world) return fail(c, PT_ERR_INVALID_ARG, "pt_gather: root >= world");
    const RoctxRange range(c, "pt_gather");
    const bool is_root = c->comm_rank == root;
    if (bytes == 0) return PT_OK;
    if (is_root ? (!recv_device && c->comm_world > 1) : !send_device) return fail(c, PT_ERR_INVALID_ARG, "pt_gather: null buffer");
    PT_HIP(c, hipSetDevice(c->device));
    Rccl& R = rccl();
    // Point-to-point over xGMI: the root posts one receive per peer, all in one group, so every inbound link of the root runs
    // at once; the root's own tiles never travel (it un-swizzles them from where they were rendered).  Stream-ordered on the
    // context's stream: after the render calls that produced `send`, before whatever the caller queues next.
    PT_NCCL(c, R.GroupStart());
    if (is_root) {
        uint32_t part = 0;
        for (uint32_t r = 0; r < c->comm_world; r++) {
            if (r == root) continue;
            const ncclResult_t e = R.Recv(static_cast<char*>(recv_device) + (size_t)part * bytes, (size_t)bytes, ncclInt8, (int)r, c->comm, c->stream);
            if (e != ncclSuccess) { (void)R.GroupEnd(); return fail(c, PT_ERR_HIP, std::string("ncclRecv: ") + R.GetErrorString(e)); }
            part++;
        }
    } else {
        const ncclResult_t e = R.Send(send_device, (size_t)bytes, ncclInt8, (int)root, c->comm, c->stream);
        if (e != ncclSuccess) { (void)R.GroupEnd(); return fail(c, PT_ERR_HIP, std::string("ncclSend: ") + R.GetErrorString(e)); }
    }
    PT_NCCL(c, R.GroupEnd());
    return PT_OK;
}

PtStatus pt_synchronize(PtContext* c)
{
    if (!c) return PT_ERR_INVALID_ARG;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, sync_all(c));
    return PT_OK;
}

}  // extern "C"
