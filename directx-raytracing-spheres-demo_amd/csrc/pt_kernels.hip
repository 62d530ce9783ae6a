// pt_kernels.hip -- the wavefront kernel set of the bounce loop for gfx950 (wave64).
//
//   primary_kernel   ray generation (Camera.hlsli:27-41, Raytracing.hlsl:106-138) + primary closest hit
//                    (GBufferGeneration.hlsl:125-129) -> ray queue 0 + hit stream
//   traverse_kernel  closest hit of every queued ray against the LBVH (replaces CastRay /
//                    RayQuery traversal, RaytracingHelpers.hlsli:57-133) -> hit stream
//   shade_kernel     one bounce of Raytracing.hlsl:213-364: environment / emission accumulate, material ->
//                    BSDFSample, lobe select + sample, PDF, BSDF, Russian roulette, throughput cutoff, sample
//                    regeneration; wave64 ballot + prefix compaction of the surviving rays into the next queue
//   trace_kernel / brute_kernel   test hooks (pt_trace_rays)
//
// Everything is compiled with -ffp-contract=off; arithmetic that decides a branch follows pt_math.h /
// pt_bsdf.h (bit-exact with the CPU oracle).  The AABB slab test is traversal-only arithmetic: it must be
// conservative, not bit-reproducible on the CPU (DESIGN.md "LBVH").
#include "pt_trace.h"

namespace pt {

// ------------------------------------------------------------------------------------------------ primary
template <bool kLds, typename StackT, bool kAlphaTex>
__global__ __launch_bounds__(kTraverseThreads) void primary_kernel(SceneView sv, PixelMap pm, FrameParams fp, RayQueue q,
                                                                   Scratch scratch, float4* __restrict__ out, FrameCounters fc)
{
    // Counter housekeeping rides on the first block (everything later in the frame is stream-ordered after this kernel):
    // fold the previous frame's ray counts into the running total, then reset the per-frame counters.
    if (blockIdx.x == 0) frame_counters_begin(fc, pm.n_slots);
    extern __shared__ float4 smem[];
    const float4* nodes = sv.nodes;
    const float4* sph = sv.sph_sorted;
    const uint32_t* ids = sv.sorted_id;
    StackT* stack;
    if (kLds) {
        stage_scene(sv, smem);
        nodes = smem;
        sph = smem + sv.n_nodes * 4u;
        ids = reinterpret_cast<const uint32_t*>(smem + sv.n_nodes * 4u + sv.n);
        stack = reinterpret_cast<StackT*>(reinterpret_cast<char*>(smem) + scene_lds_bytes(sv.n_nodes, sv.n));
    } else {
        stack = reinterpret_cast<StackT*>(smem);
    }
    stack += threadIdx.x;
    uint32_t visits[2] = { 0u, 0u };
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < pm.n_slots; slot += gridDim.x * blockDim.x) {
        const PixelRef pr = slot_to_pixel(pm, slot);
        f3 o = make_f3(0, 0, 0), d = make_f3(0, 0, 1);
        float t = kInf;
        uint32_t id = kMissId;
        uint32_t rng = 0;
        if (pr.valid) {
            float tmin, tmax;
            primary_ray(fp.cam, pr.px, pr.py, o, d, tmin, tmax);
            rng = rng_init(pr.px, pr.py, fp.frame_index);
            closest_hit_any<kLds, StackT, kAlphaTex>(sv, nodes, sph, ids, o, d, tmin, tmax, stack, blockDim.x, t, id, kLds ? nullptr : visits);
        } else if (pm.mode == 1) {
            // padding pixel of an edge tile (or a tile past the end): defined as zero
            // (out_index is always inside the packed buffer in tile mode)
            out[pr.out_index] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // invalid slots are queued as "dead": shade drops them (flags bounce = 0xFF)
        q.q0[slot] = make_float4(o.x, o.y, o.z, as_float(slot));
        q.q1[slot] = make_float4(d.x, d.y, d.z, as_float(rng));
        q.q2[slot] = make_float4(1.f, 1.f, 1.f, as_float(pr.valid ? 0u : 0xFFu));
        q.hit[slot] = make_uint2(as_uint(t), id);
        if (fp.spp > 1) scratch.primary_hit[slot] = make_uint2(as_uint(t), id);
    }
    if (!kLds) flush_visit_counters(fc.totals, visits);
}

// ------------------------------------------------------------------------------------------------ traverse
template <bool kLds, typename StackT, bool kAlphaTex>
__global__ __launch_bounds__(kTraverseThreads) void traverse_kernel(SceneView sv, RayQueue q, const uint32_t* __restrict__ count_ptr)
{
    extern __shared__ float4 smem[];
    const float4* nodes = sv.nodes;
    const float4* sph = sv.sph_sorted;
    const uint32_t* ids = sv.sorted_id;
    StackT* stack;
    const uint32_t count = *count_ptr;
    if (blockIdx.x * blockDim.x >= count) return;
    if (kLds) {
        stage_scene(sv, smem);
        nodes = smem;
        sph = smem + sv.n_nodes * 4u;
        ids = reinterpret_cast<const uint32_t*>(smem + sv.n_nodes * 4u + sv.n);
        stack = reinterpret_cast<StackT*>(reinterpret_cast<char*>(smem) + scene_lds_bytes(sv.n_nodes, sv.n));
    } else {
        stack = reinterpret_cast<StackT*>(smem);
    }
    stack += threadIdx.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float4 a = q.q0[i];
        const float4 b = q.q1[i];
        float t;
        uint32_t id;
        closest_hit_any<kLds, StackT, kAlphaTex>(sv, nodes, sph, ids, make_f3(a.x, a.y, a.z), make_f3(b.x, b.y, b.z), 0.0f, kInf, stack, blockDim.x, t, id);
        q.hit[i] = make_uint2(as_uint(t), id);
    }
}

// ------------------------------------------------------------------------------------------------ traverse, ray replacement
// Closest hit for BVHs in global memory, where per-ray node visits have a heavy tail (2^20-sphere scene: mean 101, p99 407)
// and a wave would otherwise idle until its slowest lane finishes: persistent waves pull rays from a work cursor and a
// lane that finishes its ray is handed a new one as soon as fewer than kRefillBelow lanes of its wave are busy.
constexpr uint32_t kRefillBelow = 44;

template <typename StackT, bool kWide, bool kAlphaTex>
__global__ __launch_bounds__(256) void traverse_dyn_kernel(SceneView sv, RayQueue q, const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ cursor,
                                                           unsigned long long* __restrict__ totals)
{
    uint32_t visits[2] = { 0u, 0u };  // {node visits, sphere tests} of this lane: the scene term of SURVEY 8(d)'s byte accounting
    extern __shared__ float4 smem[];
    StackT* stack = reinterpret_cast<StackT*>(smem) + threadIdx.x;
    const uint32_t stride = blockDim.x;
    const uint32_t count = *count_ptr;
    const float4* __restrict__ nodes = kWide ? sv.wide : sv.nodes;
    const float4* __restrict__ sph = sv.sph_sorted;
    const uint32_t* __restrict__ ids = sv.sorted_id;
    const uint32_t lane = lane_id();

    bool active = false, more = true;
    uint32_t idx = 0, sp = 0, best_id = kMissId;
    int node = kTraversalDone;
    f3 o = make_f3(0.f, 0.f, 0.f), d = make_f3(0.f, 0.f, 1.f);
    float ix = 0.f, iy = 0.f, iz = 0.f, ox = 0.f, oy = 0.f, oz = 0.f, best = kInf;
    for (;;) {
        const unsigned long long act = __ballot(active);
        if (more && (uint32_t)__popcll(act) < kRefillBelow) {
            const unsigned long long idle = ~act;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cursor, n_idle);
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t my = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!active && my < count) {
                const float4 a = q.q0[my], b = q.q1[my];
                o = make_f3(a.x, a.y, a.z); d = make_f3(b.x, b.y, b.z);
                ix = slab_rcp(d.x); iy = slab_rcp(d.y); iz = slab_rcp(d.z);
                ox = -o.x * ix; oy = -o.y * iy; oz = -o.z * iz;
                node = 0; sp = 0; best = kInf; best_id = kMissId;
                idx = my; active = true;
            }
            more = base + n_idle < count;
        }
        if (!__ballot(active)) break;
        bool finished = false;
        if (active) {
            uint32_t budget = sv.descent_cap;
            while (node >= 0) {
                visits[0]++;
                if (kWide) { wide_visit<StackT>(nodes, node, sp, stack, stride, ix, iy, iz, ox, oy, oz, 0.0f, best); if (--budget == 0u) break; continue; }
                const float4 n0 = nodes[node * 4 + 0];
                const float4 n1 = nodes[node * 4 + 1];
                const float4 n2 = nodes[node * 4 + 2];
                const float4 n3 = nodes[node * 4 + 3];
                float ax = pt_fma(n0.x, ix, ox), bx = pt_fma(n0.w, ix, ox);
                float ay = pt_fma(n0.y, iy, oy), by = pt_fma(n1.x, iy, oy);
                float az = pt_fma(n0.z, iz, oz), bz = pt_fma(n1.y, iz, oz);
                const float tn0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                const float tf0 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best));
                ax = pt_fma(n1.z, ix, ox); bx = pt_fma(n2.y, ix, ox);
                ay = pt_fma(n1.w, iy, oy); by = pt_fma(n2.z, iy, oy);
                az = pt_fma(n2.x, iz, oz); bz = pt_fma(n2.w, iz, oz);
                const float tn1 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                const float tf1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best));
                const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
                const int c0 = __builtin_bit_cast(int, n3.x), c1 = __builtin_bit_cast(int, n3.y);
                if (h0 && h1) {
                    const bool swap = tn1 < tn0;
                    stack[sp] = stack_encode<StackT>(swap ? c0 : c1);
                    sp += stride;
                    node = swap ? c1 : c0;
                } else if (h0) {
                    node = c0;
                } else if (h1) {
                    node = c1;
                } else if (sp == 0) {
                    node = kTraversalDone;
                } else {
                    sp -= stride;
                    node = stack_decode(stack[sp]);
                }
                if (--budget == 0u) break;
            }
            if (node == kTraversalDone) {
                finished = true;
            } else if (node < 0) {
                visits[1]++;
                const uint32_t k = ~(uint32_t)node;
                const float4 s = sph[k];
                float t;
                if (intersect_sphere(o, d, 0.0f, kInf, make_f3(s.x, s.y, s.z), s.w, t)) {
                    const uint32_t idf = ids[k];
                    if ((idf >> kIdClassShift) == 0u || alpha_candidate<kAlphaTex>(sv, idf, make_f3(s.x, s.y, s.z), s.w, o, d, t)) {
                        const uint32_t id = idf & kIdMask;
                        if (t < best || (t == best && best_id != kMissId && id < best_id)) { best = t; best_id = id; }
                    }
                }
                if (sp == 0) finished = true;
                else { sp -= stride; node = stack_decode(stack[sp]); }
            }
        }
        if (active && finished) {
            q.hit[idx] = make_uint2(as_uint(best), best_id);
            active = false;
        }
    }
    flush_visit_counters(totals, visits);
}

template <bool kTex>
__global__ __launch_bounds__(kShadeThreads) void shade_kernel(SceneView sv, PixelMap pm, FrameParams fp, RayQueue qin, RayQueue qout,
                                                              Scratch scratch, float4* __restrict__ out,
                                                              const uint32_t* __restrict__ count_in_ptr, uint32_t* __restrict__ count_out_ptr)
{
    __shared__ uint32_t s_wave_count[kShadeThreads / 64];
    __shared__ uint32_t s_block_base;
    const uint32_t count = *count_in_ptr;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        bool emit = false;
        PathState ps;
        if (i < count) {
            ps = load_path(qin, i);
            const uint2 h = qin.hit[i];
            if (ps.bounce != 0xFFu) emit = shade_step<true, kTex>(sv, pm, fp, scratch, out, ps, as_float(h.x), h.y);
        }
        // ---- wave64 ballot + prefix compaction into the next queue; one atomic per block
        const unsigned long long mask = __ballot(emit);
        const uint32_t wave_n = __popcll(mask);
        const uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave_count[wave] = wave_n;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (uint32_t w = 0; w < kShadeThreads / 64; w++) { const uint32_t c = s_wave_count[w]; s_wave_count[w] = total; total += c; }
            s_block_base = total ? atomicAdd(count_out_ptr, total) : 0u;
        }
        __syncthreads();
        if (emit) store_path(qout, s_block_base + s_wave_count[wave] + prefix, ps);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ fused tail
// When the queue has become small, per-launch latency (dispatch + BVH staging + one traversal chain) dominates a
// wavefront pass.  The tail kernel finishes every queued path in ONE launch: each lane alternates closest_hit and
// shade_step in registers until its pixel is done (persistent threads; no queue traffic, no compaction).
template <bool kLds, typename StackT, bool kTex>
__global__ __launch_bounds__(kTailThreads) void tail_kernel(SceneView sv, PixelMap pm, FrameParams fp, RayQueue qin, Scratch scratch,
                                                            float4* __restrict__ out, const uint32_t* __restrict__ count_ptr,
                                                            unsigned long long* __restrict__ tail_rays, unsigned long long* __restrict__ totals)
{
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * blockDim.x >= count) return;
    uint32_t visits[2] = { 0u, 0u };
    const float4* nodes = sv.nodes;
    const float4* sph = sv.sph_sorted;
    const uint32_t* ids = sv.sorted_id;
    StackT* stack;
    if (kLds) {
        stage_scene(sv, smem);
        nodes = smem;
        sph = smem + sv.n_nodes * 4u;
        ids = reinterpret_cast<const uint32_t*>(smem + sv.n_nodes * 4u + sv.n);
        stack = reinterpret_cast<StackT*>(reinterpret_cast<char*>(smem) + scene_lds_bytes(sv.n_nodes, sv.n));
    } else {
        stack = reinterpret_cast<StackT*>(smem);
    }
    stack += threadIdx.x;
    uint32_t my_rays = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        PathState ps = load_path(qin, i);
        if (ps.bounce == 0xFFu) continue;
        for (;;) {
            float t;
            uint32_t id;
            closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, ps.o, ps.d, 0.0f, kInf, stack, blockDim.x, t, id, kLds ? nullptr : visits);
            if (!shade_step<true, kTex>(sv, pm, fp, scratch, out, ps, t, id)) break;
            my_rays++;  // rays spawned inside the tail (the input queue's rays are already in counts[])
        }
    }
    block_atomic_add(tail_rays, my_rays);
    if (!kLds) flush_visit_counters(totals, visits);
}

// ------------------------------------------------------------------------------------------------ direct illumination (row N4)
// One thread per slot, before the bounce passes (the reference's RTXDI passes likewise run before Raytracing.hlsl and
// hand it a DI texture): re-trace the primary ray, evaluate the primary surface, sample ONE emissive sphere (uniform
// choice, uniform direction in its cone), trace the shadow ray with the ordinary closest-hit query -- the emitter must be
// the first thing it meets -- and store  DI = Le * (f_diffuse + f_specular) cos * n_lights / pdf.
// shade_step drops the emission of first-bounce hits reached through a reflective lobe and adds DI to the final radiance.  Own RNG stream; both rays are counted.
template <bool kLds, typename StackT, bool kTex>
__global__ __launch_bounds__(kTraverseThreads) void di_kernel(SceneView sv, PixelMap pm, FrameParams fp, float4* __restrict__ di, uint2* __restrict__ primary_hit,
                                                              unsigned long long* __restrict__ ray_counter)
{
    extern __shared__ float4 smem[];
    const float4* nodes = sv.nodes;
    const float4* sph = sv.sph_sorted;
    const uint32_t* ids = sv.sorted_id;
    StackT* stack;
    if (kLds) {
        stage_scene(sv, smem);
        nodes = smem;
        sph = smem + sv.n_nodes * 4u;
        ids = reinterpret_cast<const uint32_t*>(smem + sv.n_nodes * 4u + sv.n);
        stack = reinterpret_cast<StackT*>(reinterpret_cast<char*>(smem) + scene_lds_bytes(sv.n_nodes, sv.n));
    } else {
        stack = reinterpret_cast<StackT*>(smem);
    }
    stack += threadIdx.x;
    uint32_t my_rays = 0;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < pm.n_slots; slot += gridDim.x * blockDim.x) {
        const PixelRef pr = slot_to_pixel(pm, slot);
        f3 est = make_f3(0.f, 0.f, 0.f);
        if (pr.valid) {
            f3 o, d;
            float tmin, tmax, t;
            uint32_t id;
            primary_ray(fp.cam, pr.px, pr.py, o, d, tmin, tmax);
            closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, o, d, tmin, tmax, stack, blockDim.x, t, id);
            // THE primary trace of this frame: the primary pass that follows reads the hit instead of tracing again (as the
            // reference's RTXDI passes and Raytracing.hlsl both start from the G-buffer); it is counted there (queue 0)
            primary_hit[slot] = make_uint2(as_uint(t), id);
            if (id != kMissId) {
                const HitMaterial hm = hit_material<kTex>(sv, id, o, d, t, true);
                est = di_estimate<kTex>(sv, fp, pr.px, pr.py, id, d, hm,
                                        [&](f3 so, f3 sd, float& t2, uint32_t& id2) { closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, so, sd, 0.0f, kInf, stack, blockDim.x, t2, id2); },
                                        my_rays);
            }
        }
        di[slot] = make_float4(est.x, est.y, est.z, 0.0f);
    }
    block_atomic_add(ray_counter, my_rays);
}

// ------------------------------------------------------------------------------------------------ test hooks
template <bool kLds, typename StackT>
__global__ __launch_bounds__(kTraverseThreads) void trace_kernel(SceneView sv, const float* __restrict__ o, const float* __restrict__ d,
                                                                 uint32_t n_rays, float tmin, float* __restrict__ out_t, uint32_t* __restrict__ out_id,
                                                                 uint2* __restrict__ out_visits)
{
    extern __shared__ float4 smem[];
    const float4* nodes = sv.nodes;
    const float4* sph = sv.sph_sorted;
    const uint32_t* ids = sv.sorted_id;
    StackT* stack;
    if (kLds) {
        stage_scene(sv, smem);
        nodes = smem;
        sph = smem + sv.n_nodes * 4u;
        ids = reinterpret_cast<const uint32_t*>(smem + sv.n_nodes * 4u + sv.n);
        stack = reinterpret_cast<StackT*>(reinterpret_cast<char*>(smem) + scene_lds_bytes(sv.n_nodes, sv.n));
    } else {
        stack = reinterpret_cast<StackT*>(smem);
    }
    stack += threadIdx.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += gridDim.x * blockDim.x) {
        float t;
        uint32_t id;
        uint32_t v[2] = { 0, 0 };
        if (out_visits)
            if (!kLds && sv.wide)
                closest_hit<StackT, true, true, true>(sv, sv.wide, sph, ids, sv.n, make_f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), make_f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]),
                                                      tmin, kInf, stack, blockDim.x, t, id, v);
            else
                closest_hit<StackT, true, false, true>(sv, nodes, sph, ids, sv.n, make_f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), make_f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]),
                                                       tmin, kInf, stack, blockDim.x, t, id, v);
        else
            closest_hit_any<kLds, StackT, true>(sv, nodes, sph, ids, make_f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), make_f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]),
                                                tmin, kInf, stack, blockDim.x, t, id);
        out_t[i] = t;
        out_id[i] = id;
        if (out_visits) out_visits[i] = make_uint2(v[0], v[1]);
    }
}

__global__ void brute_kernel(SceneView sv, const float* __restrict__ o, const float* __restrict__ d, uint32_t n_rays, float tmin,
                             float* __restrict__ out_t, uint32_t* __restrict__ out_id)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += gridDim.x * blockDim.x) {
        const f3 oo = make_f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd = make_f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        float best = kInf;
        uint32_t best_id = kMissId;
        for (uint32_t k = 0; k < sv.n; k++) {
            const float4 s = sv.sph[k];
            float t;
            if (intersect_sphere(oo, dd, tmin, kInf, make_f3(s.x, s.y, s.z), s.w, t)) {
                const uint32_t idf = k | (sv.alpha_class ? sv.alpha_class[k] << kIdClassShift : 0u);
                if ((idf >> kIdClassShift) == 0u || alpha_candidate<true>(sv, idf, make_f3(s.x, s.y, s.z), s.w, oo, dd, t))
                    if (t < best) { best = t; best_id = k; }
            }
        }
        out_t[i] = best;
        out_id[i] = best_id;
    }
}

// Tile un-swizzle after the exchange: packed = [part][tiles][ts*ts] float4 -> frame W*H float4, where part i
// holds, in increasing tile order, the tiles t with first0 + i*run <= t % stride < first0 + (i+1)*run; consecutive parts
// are part_stride float4 apart.  Pixels of tiles owned by none of the n_parts parts are left untouched (another call
// with the other parts fills them).
// kRgb: the parts hold 3 floats per pixel (pt_pack_rgb: alpha is 1 for every pixel of the frame, so it need not travel over the
// links -- 25 % fewer bytes into the assembling GPU); part_stride is in pixels either way.
template <bool kRgb>
__global__ void unpack_tiles_kernel(const float4* __restrict__ packed, float4* __restrict__ frame, uint32_t w, uint32_t h, uint32_t ts,
                                    uint32_t tiles_x, uint32_t first0, uint32_t run, uint32_t stride, uint32_t n_parts, uint64_t part_stride)
{
    const uint64_t n = (uint64_t)w * h;  // up to 65535^2 pixels: 64-bit count and index
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(p % w), y = (uint32_t)(p / w);
        const uint32_t gt = (y / ts) * tiles_x + (x / ts);
        const uint32_t period = gt / stride, res = gt - period * stride;
        if (res < first0) continue;
        const uint32_t d = res - first0, part = d / run;
        if (part >= n_parts) continue;
        const uint32_t k = period * run + (d - part * run);
        const size_t src = (size_t)part * part_stride + (size_t)k * ts * ts + (y % ts) * ts + (x % ts);
        if (kRgb) {
            const float* q = reinterpret_cast<const float*>(packed) + 3u * src;
            frame[p] = make_float4(q[0], q[1], q[2], 1.0f);
        } else {
            frame[p] = packed[src];
        }
    }
}

// float4 (r, g, b, a) -> 3 floats per pixel, for the exchange buffers
__global__ void pack_rgb_kernel(const float4* __restrict__ src, float* __restrict__ dst, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        dst[3u * i] = v.x; dst[3u * i + 1u] = v.y; dst[3u * i + 2u] = v.z;
    }
}

// Row N3: display transform (20 B / pixel: HBM-bound) and progressive accumulation (48 B / pixel).
__global__ void tonemap_kernel(const float4* __restrict__ hdr, uint32_t* __restrict__ out, uint32_t n, PtToneMapParams p)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 c = hdr[i];
        out[i] = tonemap_pixel(make_f3(c.x, c.y, c.z), p);
    }
}

__global__ void accumulate_kernel(float4* __restrict__ accum, const float4* __restrict__ rad, uint32_t n, float inv, int first)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 x = rad[i];
        float4 a = first ? x : accum[i];
        if (!first) {
            a.x = accumulate_value(a.x, x.x, inv, false); a.y = accumulate_value(a.y, x.y, inv, false);
            a.z = accumulate_value(a.z, x.z, inv, false); a.w = accumulate_value(a.w, x.w, inv, false);
        }
        accum[i] = a;
    }
}

// totals[0] += sum of counts[1..n_iters] (secondary rays of this frame); one thread
// Fold one parity's counters into the running total (pt_get_totals / stats); leaves them zeroed.
__global__ void flush_counters_kernel(uint32_t* counts, uint32_t n_counts, unsigned long long* tail, unsigned long long* totals, uint32_t* host_counts)
{
    if (blockIdx.x == 0) fold_counters(counts, n_counts, tail, totals, host_counts);
}

// bit 31 of every material's AlphaMode word (device copy) := the sphere has texture maps (tex_maps[i * 8 + 7]; null = no sphere has)
__global__ void material_map_flag_kernel(uint32_t* __restrict__ mats_words, const uint32_t* __restrict__ tex_maps, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t w = mats_words[(size_t)i * 16u + 12u] & ~kMaterialHasMaps;
        if (tex_maps && tex_maps[(size_t)i * 8u + 7u]) w |= kMaterialHasMaps;
        mats_words[(size_t)i * 16u + 12u] = w;
    }
}

// leaf ids the traversal reads: Morton order -> original id | alpha class << 30 (pt_device.h)
__global__ void leaf_ids_kernel(const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ alpha_class, uint32_t n, uint32_t* __restrict__ out)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t id = sorted_id[k];
        out[k] = id | (alpha_class[id] << kIdClassShift);
    }
}

// ------------------------------------------------------------------------------------------------ launch wrappers
static uint32_t traverse_lds_bytes(const SceneView& sv, uint32_t stack_elem)
{
    const uint32_t stack = traverse_threads(sv.lds_scene != 0) * sv.stack_depth * stack_elem;
    return (sv.lds_scene ? scene_lds_bytes(sv.n_nodes, sv.n) : 0u) + stack;
}

uint32_t traverse_lds_bytes_for(uint32_t n_nodes, uint32_t n, uint32_t depth, bool lds_scene, uint32_t threads)
{
    const uint32_t elem = n_nodes < 32767u ? 2u : 4u;
    if (threads == 0) threads = traverse_threads(lds_scene);
    return (lds_scene ? scene_lds_bytes(n_nodes, n) : 0u) + threads * depth * elem;
}

// KERNEL<kLds, StackT, ...>: EXTRA = further template arguments (with their leading comma) or nothing
#define PT_DISPATCH_TRAVERSE(KERNEL, EXTRA, GRID, STREAM, ...)                                                 \
    do {                                                                                                        \
        const bool small = sv.n_nodes < 32767u;                                                                 \
        const uint32_t lds = traverse_lds_bytes(sv, small ? 2u : 4u);                                           \
        if (lds + kStaticLdsMargin > 65536u) {                                                                                     \
            const void* fn = sv.lds_scene ? (small ? (const void*)KERNEL<true, uint16_t EXTRA> : (const void*)KERNEL<true, uint32_t EXTRA>)   \
                                          : (small ? (const void*)KERNEL<false, uint16_t EXTRA> : (const void*)KERNEL<false, uint32_t EXTRA>); \
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        }                                                                                                       \
        if (sv.lds_scene) {                                                                                     \
            if (small) hipLaunchKernelGGL((KERNEL<true, uint16_t EXTRA>), dim3(GRID), dim3(traverse_threads(true)), lds, STREAM, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<true, uint32_t EXTRA>), dim3(GRID), dim3(traverse_threads(true)), lds, STREAM, __VA_ARGS__);       \
        } else {                                                                                                \
            if (small) hipLaunchKernelGGL((KERNEL<false, uint16_t EXTRA>), dim3(GRID), dim3(traverse_threads(false)), lds, STREAM, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<false, uint32_t EXTRA>), dim3(GRID), dim3(traverse_threads(false)), lds, STREAM, __VA_ARGS__);       \
        }                                                                                                       \
    } while (0)
#define PT_COMMA_TRUE , true
#define PT_COMMA_FALSE , false

hipError_t launch_primary(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& q, const Scratch& scratch,
                          float4* out, const FrameCounters& fc, uint32_t grid, hipStream_t stream)
{
    // (the kAlphaTex form only when some sphere's hits are alpha-tested against a map)
    if (sv.alpha_tested) PT_DISPATCH_TRAVERSE(primary_kernel, PT_COMMA_TRUE, grid, stream, sv, pm, fp, q, scratch, out, fc);
    else PT_DISPATCH_TRAVERSE(primary_kernel, PT_COMMA_FALSE, grid, stream, sv, pm, fp, q, scratch, out, fc);
    return hipGetLastError();
}

hipError_t launch_traverse(const SceneView& sv, const RayQueue& q, const uint32_t* count_ptr, uint32_t grid, hipStream_t stream)
{
    if (sv.alpha_tested) PT_DISPATCH_TRAVERSE(traverse_kernel, PT_COMMA_TRUE, grid, stream, sv, q, count_ptr);
    else PT_DISPATCH_TRAVERSE(traverse_kernel, PT_COMMA_FALSE, grid, stream, sv, q, count_ptr);
    return hipGetLastError();
}

hipError_t launch_traverse_dyn(const SceneView& sv, const RayQueue& q, const uint32_t* count_ptr, uint32_t* cursor, unsigned long long* totals, uint32_t grid,
                               hipStream_t stream)
{
    const bool small = sv.n_nodes < 32767u;
    const uint32_t lds = 256u * sv.stack_depth * (small ? 2u : 4u);
#define PT_DYN(T, W, A)                                                                                                          \
    do {                                                                                                                          \
        if (lds + kStaticLdsMargin > 65536u) (void)hipFuncSetAttribute((const void*)traverse_dyn_kernel<T, W, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((traverse_dyn_kernel<T, W, A>), dim3(grid), dim3(256), lds, stream, sv, q, count_ptr, cursor, totals);  \
    } while (0)
#define PT_DYN2(T, W) do { if (sv.alpha_tested) PT_DYN(T, W, true); else PT_DYN(T, W, false); } while (0)
    if (sv.wide) { if (small) PT_DYN2(uint16_t, true); else PT_DYN2(uint32_t, true); }
    else { if (small) PT_DYN2(uint16_t, false); else PT_DYN2(uint32_t, false); }
#undef PT_DYN2
#undef PT_DYN
    return hipGetLastError();
}

hipError_t launch_beams(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, float slack, uint32_t* lists, hipStream_t stream)
{
    const uint32_t n_blocks = pm.n_slots >> 6;
    if (n_blocks == 0) return hipSuccess;
    const bool small = sv.n_nodes < 32767u;
    const uint32_t stack_bytes = 256u * (sv.stack_depth + 1u) * (small ? 2u : 4u);
    const uint32_t lds = (sv.lds_scene ? scene_lds_bytes(sv.n_nodes, sv.n) : 0u) + stack_bytes;
    const uint32_t grid = (n_blocks + 255u) / 256u;
#define PT_BEAM(L, T)                                                                                                                            \
    do {                                                                                                                                          \
        if (lds + kStaticLdsMargin > 65536u) (void)hipFuncSetAttribute((const void*)beam_kernel<L, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((beam_kernel<L, T>), dim3(grid), dim3(256), lds, stream, sv, pm, fp, slack, lists);                                           \
    } while (0)
    if (sv.lds_scene) { if (small) PT_BEAM(true, uint16_t); else PT_BEAM(true, uint32_t); }
    else { if (small) PT_BEAM(false, uint16_t); else PT_BEAM(false, uint32_t); }
#undef PT_BEAM
    return hipGetLastError();
}

hipError_t launch_tail(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const Scratch& scratch, float4* out,
                       const uint32_t* count_ptr, unsigned long long* tail_rays, unsigned long long* totals, uint32_t grid, hipStream_t stream)
{
    const bool small = sv.n_nodes < 32767u;
    const uint32_t elem = small ? 2u : 4u;
    const uint32_t lds = (sv.lds_scene ? scene_lds_bytes(sv.n_nodes, sv.n) : 0u) + kTailThreads * sv.stack_depth * elem;
#define PT_TAIL2(L, T, X)                                                                                                  \
    do {                                                                                                                    \
        if (lds + kStaticLdsMargin > 65536u) (void)hipFuncSetAttribute((const void*)tail_kernel<L, T, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((tail_kernel<L, T, X>), dim3(grid), dim3(kTailThreads), lds, stream, sv, pm, fp, qin, scratch, out, count_ptr, tail_rays, totals); \
    } while (0)
#define PT_TAIL(L, T) do { if (sv.tex_maps) PT_TAIL2(L, T, true); else PT_TAIL2(L, T, false); } while (0)
    if (sv.lds_scene) { if (small) PT_TAIL(true, uint16_t); else PT_TAIL(true, uint32_t); }
    else { if (small) PT_TAIL(false, uint16_t); else PT_TAIL(false, uint32_t); }
#undef PT_TAIL
#undef PT_TAIL2
    return hipGetLastError();
}

hipError_t launch_bounce(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const RayQueue& qout,
                         const Scratch& scratch, float4* out, const uint32_t* count_in, uint32_t* count_out, const FrameCounters& fc,
                         bool primary, bool loop, bool inline2, uint32_t threads, uint32_t grid, hipStream_t stream)
{
    const bool small = sv.n_nodes < 32767u;
    if (sv.lds_scene) {
        if (!small) return hipErrorInvalidValue;  // (an LDS-resident tree cannot be this large: pt_build_accel's 64 KB budget)
        return launch_bounce_lds16(sv, pm, fp, qin, qout, scratch, out, count_in, count_out, fc, primary, loop, inline2, threads, grid, stream);
    }
    if (small) return launch_bounce_g16(sv, pm, fp, qin, qout, scratch, out, count_in, count_out, fc, primary, loop, inline2, threads, grid, stream);
    return launch_bounce_g32(sv, pm, fp, qin, qout, scratch, out, count_in, count_out, fc, primary, loop, inline2, threads, grid, stream);
}

hipError_t launch_shade(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const RayQueue& qout,
                        const Scratch& scratch, float4* out, const uint32_t* count_in, uint32_t* count_out, uint32_t grid, hipStream_t stream)
{
    if (sv.tex_maps) hipLaunchKernelGGL(shade_kernel<true>, dim3(grid), dim3(kShadeThreads), 0, stream, sv, pm, fp, qin, qout, scratch, out, count_in, count_out);
    else hipLaunchKernelGGL(shade_kernel<false>, dim3(grid), dim3(kShadeThreads), 0, stream, sv, pm, fp, qin, qout, scratch, out, count_in, count_out);
    return hipGetLastError();
}

hipError_t launch_trace(const SceneView& sv, const float* o, const float* d, uint32_t n_rays, float tmin, int use_bvh, float* out_t,
                        uint32_t* out_id, uint2* out_visits, hipStream_t stream)
{
    const uint32_t tt = traverse_threads(sv.lds_scene != 0);
    const uint32_t grid = (n_rays + tt - 1) / tt < 2048u ? (n_rays + tt - 1) / tt : 2048u;
    if (grid == 0) return hipSuccess;
    if (use_bvh) {
        PT_DISPATCH_TRAVERSE(trace_kernel, , grid, stream, sv, o, d, n_rays, tmin, out_t, out_id, out_visits);
    } else {
        hipLaunchKernelGGL(brute_kernel, dim3(grid), dim3(kTraverseThreads), 0, stream, sv, o, d, n_rays, tmin, out_t, out_id);
    }
    return hipGetLastError();
}

hipError_t launch_flush_counters(uint32_t* counts, uint32_t n_counts, unsigned long long* tail, unsigned long long* totals, uint32_t* host_counts,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(flush_counters_kernel, dim3(1), dim3(256), 0, stream, counts, n_counts, tail, totals, host_counts);
    return hipGetLastError();
}

hipError_t launch_unpack_tiles(const float4* packed, float4* frame, uint32_t w, uint32_t h, uint32_t ts, uint32_t tiles_x, uint32_t first0,
                               uint32_t run, uint32_t stride, uint32_t n_parts, uint64_t part_stride, bool rgb,
                               hipStream_t stream)
{
    const uint64_t n = (uint64_t)w * h;
    const uint32_t grid = (uint32_t)((n + 255u) / 256u < 4096u ? (n + 255u) / 256u : 4096u);
    if (rgb) hipLaunchKernelGGL(unpack_tiles_kernel<true>, dim3(grid), dim3(256), 0, stream, packed, frame, w, h, ts, tiles_x, first0, run, stride, n_parts, part_stride);
    else hipLaunchKernelGGL(unpack_tiles_kernel<false>, dim3(grid), dim3(256), 0, stream, packed, frame, w, h, ts, tiles_x, first0, run, stride, n_parts, part_stride);
    return hipGetLastError();
}

hipError_t launch_di(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, float4* di, uint2* primary_hit, unsigned long long* ray_counter, uint32_t grid,
                     hipStream_t stream)
{
    const bool small = sv.n_nodes < 32767u;
    const uint32_t threads = traverse_threads(sv.lds_scene != 0);
    const uint32_t lds = traverse_lds_bytes(sv, small ? 2u : 4u);
#define PT_DI2(L, T, X)                                                                                                     \
    do {                                                                                                                    \
        if (lds + kStaticLdsMargin > 65536u) (void)hipFuncSetAttribute((const void*)di_kernel<L, T, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((di_kernel<L, T, X>), dim3(grid), dim3(threads), lds, stream, sv, pm, fp, di, primary_hit, ray_counter);       \
    } while (0)
#define PT_DI(L, T) do { if (sv.tex_maps) PT_DI2(L, T, true); else PT_DI2(L, T, false); } while (0)
    if (sv.lds_scene) { if (small) PT_DI(true, uint16_t); else PT_DI(true, uint32_t); }
    else { if (small) PT_DI(false, uint16_t); else PT_DI(false, uint32_t); }
#undef PT_DI
#undef PT_DI2
    return hipGetLastError();
}

hipError_t launch_material_map_flags(float4* mats, const uint32_t* tex_maps, uint32_t n, hipStream_t stream)
{
    const uint32_t grid = (n + 255u) / 256u < 4096u ? (n + 255u) / 256u : 4096u;
    hipLaunchKernelGGL(material_map_flag_kernel, dim3(grid ? grid : 1u), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(mats), tex_maps, n);
    return hipGetLastError();
}

hipError_t launch_leaf_ids(const uint32_t* sorted_id, const uint32_t* alpha_class, uint32_t n, uint32_t* out, hipStream_t stream)
{
    const uint32_t grid = (n + 255u) / 256u < 4096u ? (n + 255u) / 256u : 4096u;
    hipLaunchKernelGGL(leaf_ids_kernel, dim3(grid ? grid : 1u), dim3(256), 0, stream, sorted_id, alpha_class, n, out);
    return hipGetLastError();
}

hipError_t launch_pack_rgb(const float4* src, float* dst, uint64_t n, hipStream_t stream)
{
    const uint64_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(pack_rgb_kernel, dim3((uint32_t)(blocks < 8192u ? (blocks ? blocks : 1u) : 8192u)), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float4* hdr, uint32_t* out, uint32_t n, const PtToneMapParams& p, hipStream_t stream)
{
    const uint32_t grid = (n + 255u) / 256u < 8192u ? (n + 255u) / 256u : 8192u;
    hipLaunchKernelGGL(tonemap_kernel, dim3(grid ? grid : 1u), dim3(256), 0, stream, hdr, out, n, p);
    return hipGetLastError();
}

hipError_t launch_accumulate(float4* accum, const float4* rad, uint32_t n, uint32_t frames_accumulated, hipStream_t stream)
{
    const uint32_t grid = (n + 255u) / 256u < 8192u ? (n + 255u) / 256u : 8192u;
    const float inv = 1.0f / (float)(frames_accumulated + 1u);
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid ? grid : 1u), dim3(256), 0, stream, accum, rad, n, inv, frames_accumulated == 0 ? 1 : 0);
    return hipGetLastError();
}

}  // namespace pt
