// pt_trace.h -- device code shared by the kernel translation units: BVH traversal (closest_hit, wide_visit, primary-beam
// helpers), the shading step of the bounce loop (shade_step: Raytracing.hlsl:213-364) and the fused trace + shade kernel
// bounce_kernel with its launcher template.  pt_kernels.hip holds the other kernels; pt_bounce_*.hip instantiate bounce_kernel
// for one (BVH residence, stack entry type) pair each, so that the ~100 instances compile in parallel.
//
// Everything is compiled with -ffp-contract=off; arithmetic that decides a branch follows pt_math.h /
// pt_bsdf.h (bit-exact with the CPU oracle).  The AABB slab test is traversal-only arithmetic: it must be
// conservative, not bit-reproducible on the CPU (DESIGN.md "LBVH").
#pragma once

#include "pt_kernels.h"

namespace pt {

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Reciprocal of a ray-direction component for the slab test t = fma(plane, inv, -o * inv).  A zero (or denormal-tiny)
// component would give inv = inf and then inf - inf = NaN for the planes on one side of the origin, and fmax(-inf, NaN)
// = -inf would wrongly close the slab: the component is replaced by +-1e-30, which keeps every product finite (scene
// coordinates are far below 1e8) and makes the slab interval (-huge, +huge) when the origin lies between the planes and
// empty otherwise -- the exact behaviour of an axis-parallel ray.  Found by tests/test_gpu_fuzz.py (centre column of
// an odd-width frame with zero jitter: d.x == 0).
__device__ __forceinline__ float slab_rcp(float d)
{
    const float kTiny = 1e-30f;
    return fast_rcp(__builtin_fabsf(d) < kTiny ? __builtin_copysignf(kTiny, d) : d);
}

// Adds the sum of `v` over the workgroup to *counter with ONE atomic (wave shuffle -> LDS -> thread 0).  Same-address
// device-scope atomics from all 8 XCDs serialise; one per wave has been measured to dominate short kernels.
__device__ __forceinline__ void block_atomic_add(unsigned long long* counter, unsigned long long v)
{
    __shared__ unsigned long long s_part[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (uint32_t w = 0; w < (blockDim.x + 63u) / 64u; w++) sum += s_part[w];
        if (sum) atomicAdd(counter, sum);
    }
}

// Called by every thread of ONE block: fold a finished frame's counters (queue sizes 1..n + its tail counter) into
// the running totals and leave them zeroed.
__device__ __forceinline__ void fold_counters(uint32_t* __restrict__ counts, uint32_t n_counts, unsigned long long* __restrict__ tail,
                                              unsigned long long* __restrict__ totals, uint32_t* __restrict__ host_counts)
{
    __shared__ unsigned long long s_sum[16];
    unsigned long long s = 0;
    for (uint32_t k = 1 + threadIdx.x; k <= n_counts; k += blockDim.x) s += counts[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = *tail;
        for (uint32_t w = 0; w < (blockDim.x + 63u) / 64u; w++) t += s_sum[w];
        totals[0] += t;
        totals[1] = t;
        *tail = 0ull;
    }
    __syncthreads();
    // publish the folded frame's queue sizes to the host (it sizes later frames' grids from them); counts[0] == 0 means
    // these counters hold no frame (already folded)
    if (host_counts && counts[0] != 0u)
        for (uint32_t k = threadIdx.x; k <= n_counts; k += blockDim.x) host_counts[k] = counts[k];
    for (uint32_t k = threadIdx.x; k < 2u * (n_counts + 1u); k += blockDim.x) counts[k] = 0u;  // queue sizes + work cursors
}

// First kernel of a frame, block 0: fold the previous frame (other parity), publish this frame's queue-0 size.
// This frame's counters were zeroed when the frame before the previous one was folded (or at allocation).
__device__ __forceinline__ void frame_counters_begin(const FrameCounters& fc, uint32_t n_slots)
{
    fold_counters(fc.fold_counts, fc.n_counts, fc.fold_tail, fc.totals, fc.host_counts);
    if (threadIdx.x == 0) fc.counts[0] = n_slots;
}

// Stage the BVH (nodes, Morton-ordered spheres, ids) into LDS.  Layout: [nodes | spheres | ids].
__device__ __forceinline__ void stage_scene(const SceneView& sv, float4* lds)
{
    const uint32_t n_vec = sv.n_nodes * 4u + sv.n;
    for (uint32_t i = threadIdx.x; i < n_vec; i += blockDim.x)
        lds[i] = i < sv.n_nodes * 4u ? sv.nodes[i] : sv.sph_sorted[i - sv.n_nodes * 4u];
    uint32_t* ids = reinterpret_cast<uint32_t*>(lds + n_vec);
    for (uint32_t i = threadIdx.x; i < sv.n; i += blockDim.x) ids[i] = sv.sorted_id[i];
    __syncthreads();
}

__host__ __device__ inline uint32_t scene_lds_bytes(uint32_t n_nodes, uint32_t n) { return (n_nodes * 4u + n) * 16u + ((n * 4u + 15u) & ~15u); }

// Stack entries hold a child reference: internal node index i >= 0, or leaf (Morton-sorted sphere index k) as ~k.
// The 16-bit stack (trees with < 32768 leaves) stores a leaf as 0x8000 | k.
template <typename StackT> __device__ __forceinline__ StackT stack_encode(int c);
template <> __device__ __forceinline__ uint16_t stack_encode<uint16_t>(int c) { return c < 0 ? (uint16_t)(0x8000u | (uint32_t)~c) : (uint16_t)c; }
template <> __device__ __forceinline__ uint32_t stack_encode<uint32_t>(int c) { return (uint32_t)c; }
__device__ __forceinline__ int stack_decode(uint16_t v) { return (v & 0x8000u) ? ~(int)(v & 0x7FFFu) : (int)v; }
__device__ __forceinline__ int stack_decode(uint32_t v) { return (int)v; }

constexpr int kTraversalDone = (int)0x80000000;  // not a valid leaf code (leaf codes are >= -2^30)

// fminf / fmaxf against a value the compiler cannot prove quiet (tmin and the running best cross basic blocks) cost an extra
// v_max x, x per operand and per node visit.  Neither is ever a signalling NaN, so the slab test names the instruction itself:
// v_min_f32 / v_max_f32 return the other operand for a quiet NaN exactly as fminf / fmaxf do ("Conservativeness").  Measured: C2 -0.8 %.
__device__ __forceinline__ float raw_minf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_maxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// One visit of a 4-wide node (pt_lbvh_gpu.hip collapse4_kernel: the grandchildren of an even-depth binary node, their boxes as
// 8-bit offsets on a per-node power-of-two grid; global-memory scenes): four slab tests from ONE 64-byte record, the hit
// children ordered near to far by a 5-exchange network, the nearest descended into, the others pushed far-first.  Half the
// dependent fetches AND half the sectors per ray of the binary walk (DESIGN.md "Wide nodes").  A plane's ray parameter is
// t = fma(byte, cell * (1/d), fma(origin, 1/d, -o/d)): the decode fma(byte, cell, origin) folded into the slab test.
// Sets node = kTraversalDone when nothing is left.
template <typename StackT>
__device__ __forceinline__ void wide_visit(const float4* __restrict__ wide, int& node, uint32_t& sp, StackT* stack, uint32_t stride, float ix, float iy,
                                           float iz, float ox, float oy, float oz, float tmin, float best)
{
    const uint4* __restrict__ w = reinterpret_cast<const uint4*>(wide) + (size_t)node * 4u;
    const uint4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    const float sx = as_float((w0.w & 0xFFu) << 23) * ix, sy = as_float(((w0.w >> 8) & 0xFFu) << 23) * iy, sz = as_float(((w0.w >> 16) & 0xFFu) << 23) * iz;
    const float bx0 = pt_fma(as_float(w0.x), ix, ox), by0 = pt_fma(as_float(w0.y), iy, oy), bz0 = pt_fma(as_float(w0.z), iz, oz);
    float tn[4];
    int ref[4] = { (int)w2.z, (int)w2.w, (int)w3.x, (int)w3.y };
#define PT_WIDE_CHILD(c)                                                                                                 \
    {                                                                                                                    \
        const float ax = pt_fma((float)((w1.x >> (8 * c)) & 0xFFu), sx, bx0), bx = pt_fma((float)((w1.w >> (8 * c)) & 0xFFu), sx, bx0);   \
        const float ay = pt_fma((float)((w1.y >> (8 * c)) & 0xFFu), sy, by0), by = pt_fma((float)((w2.x >> (8 * c)) & 0xFFu), sy, by0);   \
        const float az = pt_fma((float)((w1.z >> (8 * c)) & 0xFFu), sz, bz0), bz = pt_fma((float)((w2.y >> (8 * c)) & 0xFFu), sz, bz0);   \
        const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), raw_maxf(fminf(az, bz), tmin));                   \
        const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), raw_minf(fmaxf(az, bz), best));                    \
        tn[c] = (tnear <= tfar && ref[c] != kTraversalDone) ? tnear : kInf;                                             \
    }
    PT_WIDE_CHILD(0) PT_WIDE_CHILD(1) PT_WIDE_CHILD(2) PT_WIDE_CHILD(3)
#undef PT_WIDE_CHILD
#define PT_WIDE_CSWAP(a, b)                                                                                              \
    {                                                                                                                    \
        const bool sw = tn[b] < tn[a];                                                                                   \
        const float ta = sw ? tn[b] : tn[a], tb = sw ? tn[a] : tn[b];                                                    \
        const int ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];                                                  \
        tn[a] = ta; tn[b] = tb; ref[a] = ra; ref[b] = rb;                                                                \
    }
    PT_WIDE_CSWAP(0, 1) PT_WIDE_CSWAP(2, 3) PT_WIDE_CSWAP(0, 2) PT_WIDE_CSWAP(1, 3) PT_WIDE_CSWAP(1, 2)
#undef PT_WIDE_CSWAP
    if (tn[3] < kInf) { stack[sp] = stack_encode<StackT>(ref[3]); sp += stride; }
    if (tn[2] < kInf) { stack[sp] = stack_encode<StackT>(ref[2]); sp += stride; }
    if (tn[1] < kInf) { stack[sp] = stack_encode<StackT>(ref[1]); sp += stride; }
    if (tn[0] < kInf) {
        node = ref[0];
    } else if (sp == 0) {
        node = kTraversalDone;
    } else {
        sp -= stride;
        node = stack_decode(stack[sp]);
    }
}

// ------------------------------------------------------------------------------------------------ alpha-tested hits (spec S10)
// texture coordinates of the point of sphere `id` whose outward world-space normal is N (spec S6): q = the object's rotation,
// n_mesh = the mesh-space normal the coordinates (and the tangent) derive from
__device__ __forceinline__ f2 hit_uv_rot(const float4* __restrict__ rot, uint32_t id, f3 N, float4& q, f3& n_mesh)
{
    q = rot[id];
    const f3 n_obj = quat_rotate(-q.x, -q.y, -q.z, q.w, N);  // world -> object: the conjugate rotation
    // ObjectToWorld = diag(1, 1, -1) * pose (Scene.ixx:197-199): the mesh-space normal is the z mirror of the object-space
    // one (settled against the reference's screenshot with its own Earth map: without it the continents are mirrored)
    n_mesh = make_f3(n_obj.x, n_obj.y, -n_obj.z);
    return sphere_uv(n_mesh);
}
__device__ __forceinline__ f2 hit_uv(const SceneView& sv, uint32_t id, f3 N, float4& q, f3& n_mesh) { return hit_uv_rot(sv.rot, id, N, q, n_mesh); }

// IsOpaque (ShadingHelpers.hlsli:105-115) for the crossing at parameter t of a kAlphaTested sphere: alpha = BaseColor.a * the
// base-colour map's alpha at the crossing's texture coordinates (the class says EvaluateBaseColor samples, :61-72); accepted iff
// alpha >= AlphaCutoff.  Out of line (plain pointers, nothing of the kernel's argument structs): a rare path that the traversal loops
// of every textured kernel would otherwise carry several inlined copies of -- those kernels are bound by their code size as it is.
__device__ __attribute__((noinline)) inline bool crossing_is_opaque_ptr(const float4* __restrict__ mats, const uint32_t* __restrict__ tex_maps, const TexView* __restrict__ tex,
                                                                        const float4* __restrict__ rot, uint32_t id, f3 C, f3 o, f3 d, float t)
{
    const float base_alpha = mats[id * 4 + 0].w, cutoff = mats[id * 4 + 3].y;
    const uint32_t map = tex_maps[(size_t)id * 8u + kMapBaseColor];
    const f3 N = normalize(mad(t, d, o) - C);  // the hit frame's normal
    float4 q;
    f3 n_mesh;
    const f2 uv = hit_uv_rot(rot, id, N, q, n_mesh);
    float s[4];
    sample_bilinear(tex[map], uv, s);
    return base_alpha * s[3] >= cutoff;
}
__device__ __forceinline__ bool crossing_is_opaque(const SceneView& sv, uint32_t id, f3 C, f3 o, f3 d, float t)
{
    return crossing_is_opaque_ptr(sv.mats, sv.tex_maps, sv.tex, sv.rot, id, C, o, d, t);
}

// A sphere test succeeded at t for the leaf with flagged id `idf` whose class bits are set (rare): does the sphere offer an
// accepted crossing?  kAlphaInvisible: never.  kAlphaTested: the crossing at t, else the one behind it ("the next crossing" =
// intersect_sphere with tmin = the rejected t: the far root, seen from inside); t returns the accepted one.  kAlphaTex = false
// (kernels of scenes without textures, where the class cannot occur): accepted.
template <bool kAlphaTex>
__device__ __forceinline__ bool alpha_candidate(const SceneView& sv, uint32_t idf, f3 C, float r, f3 o, f3 d, float& t)
{
    if ((idf >> kIdClassShift) == kAlphaInvisible) return false;
    if (!kAlphaTex) return true;
    const uint32_t id = idf & kIdMask;
    for (;;) {
        if (crossing_is_opaque(sv, id, C, o, d, t)) return true;
        float t2;
        if (!intersect_sphere(o, d, t, kInf, C, r, t2)) return false;
        t = t2;
    }
}

// Closest hit over the LBVH ("while-while" traversal: descend internal nodes until every lane of the wave holds a
// leaf or has finished, then run the sphere tests together).  nodes/sph/ids may live in LDS or global memory (the
// address space is inferred after inlining).  stack: per-lane stack, entry e of lane l at stack[e * stride + l].
// The result is identical to brute force: nearest t, ties -> lowest original id (leaf boxes are padded so the slab
// test is conservative w.r.t. intersect_sphere; culling is <=).
// kWide: `nodes` is the 4-wide view of the tree (128-byte records, wide_visit) instead of the binary records.
// kAlphaTex: the per-crossing alpha test of kAlphaTested spheres is compiled in (alpha_candidate).
template <typename StackT, bool kCount = false, bool kWide = false, bool kAlphaTex = false>
__device__ __forceinline__ void closest_hit(const SceneView& sv, const float4* __restrict__ nodes, const float4* __restrict__ sph,
                                            const uint32_t* __restrict__ ids, uint32_t n, f3 o, f3 d, float tmin, float tmax,
                                            StackT* stack, uint32_t stride, float& t_out, uint32_t& id_out, uint32_t* visits = nullptr, uint32_t descent_cap = 0)
{
    uint32_t n_nodes_visited = 0, n_spheres_tested = 0;  // kCount only (pt_trace_rays statistics hook)
    float best = tmax;
    uint32_t best_id = kMissId;
    if (n == 1) {
        float4 s = sph[0];
        float t;
        if (intersect_sphere(o, d, tmin, kInf, make_f3(s.x, s.y, s.z), s.w, t)) {
            const uint32_t idf = ids[0];
            if ((idf >> kIdClassShift) == 0u || alpha_candidate<kAlphaTex>(sv, idf, make_f3(s.x, s.y, s.z), s.w, o, d, t))
                if (t < best) { best = t; best_id = idf & kIdMask; }
        }
        t_out = best; id_out = best_id;
        return;
    }
    const float ix = slab_rcp(d.x), iy = slab_rcp(d.y), iz = slab_rcp(d.z);
    const float ox = -o.x * ix, oy = -o.y * iy, oz = -o.z * iz;
    int node = 0;
    uint32_t sp = 0;  // stack offset in elements: a multiple of `stride` (entry e of this lane lives at stack[e * stride])
    for (;;) {
        // descent_cap > 0 bounds the node visits a lane makes before the wave turns to the sphere tests (incoherent rays in big
        // scenes: lanes that already hold a leaf otherwise idle until the longest descent of the wave ends); 0 = unbounded
        uint32_t budget = descent_cap;
        while (node >= 0) {
            if (kCount) n_nodes_visited++;
            if (kWide) { wide_visit<StackT>(nodes, node, sp, stack, stride, ix, iy, iz, ox, oy, oz, tmin, best); if (--budget == 0u) break; continue; }
            const float4 n0 = nodes[node * 4 + 0];
            const float4 n1 = nodes[node * 4 + 1];
            const float4 n2 = nodes[node * 4 + 2];
            const float4 n3 = nodes[node * 4 + 3];
            // child 0: lo = (n0.x,n0.y,n0.z) hi = (n0.w,n1.x,n1.y); child 1: lo = (n1.z,n1.w,n2.x) hi = (n2.y,n2.z,n2.w)
            // A min / max costs a whole 4-cycle slot unless its neighbour is an fma (tools/experiments/int_rate.hip, profiles/r02_valu_mix.txt):
            // child 1's plane distances are written between child 0's min / max (the scheduler has the last word; it made no measurable difference).
            const float ax = pt_fma(n0.x, ix, ox), bx = pt_fma(n0.w, ix, ox);
            const float ay = pt_fma(n0.y, iy, oy), by = pt_fma(n1.x, iy, oy);
            const float az = pt_fma(n0.z, iz, oz), bz = pt_fma(n1.y, iz, oz);
            const float lx0 = fminf(ax, bx); const float cx = pt_fma(n1.z, ix, ox);
            const float ly0 = fminf(ay, by); const float dx = pt_fma(n2.y, ix, ox);
            const float lz0 = fminf(az, bz); const float cy = pt_fma(n1.w, iy, oy);
            const float hx0 = fmaxf(ax, bx); const float dy = pt_fma(n2.z, iy, oy);
            const float hy0 = fmaxf(ay, by); const float cz = pt_fma(n2.x, iz, oz);
            const float hz0 = fmaxf(az, bz); const float dz = pt_fma(n2.w, iz, oz);
            const float tn0 = fmaxf(fmaxf(lx0, ly0), raw_maxf(lz0, tmin));
            const float tf0 = fminf(fminf(hx0, hy0), raw_minf(hz0, best));
            const float tn1 = fmaxf(fmaxf(fminf(cx, dx), fminf(cy, dy)), raw_maxf(fminf(cz, dz), tmin));
            const float tf1 = fminf(fminf(fmaxf(cx, dx), fmaxf(cy, dy)), raw_minf(fmaxf(cz, dz), best));
            const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
            const int c0 = __builtin_bit_cast(int, n3.x), c1 = __builtin_bit_cast(int, n3.y);
            if (h0 && h1) {
                const bool swap = tn1 < tn0;
                const int near_c = swap ? c1 : c0, far_c = swap ? c0 : c1;
                stack[sp] = stack_encode<StackT>(far_c);
                sp += stride;
                node = near_c;
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
        if (node == kTraversalDone) break;
        if (node >= 0) continue;  // the budget ran out mid-descent
        {
            if (kCount) n_spheres_tested++;
            const uint32_t k = ~(uint32_t)node;
            const float4 s = sph[k];
            float t;
            if (intersect_sphere(o, d, tmin, kInf, make_f3(s.x, s.y, s.z), s.w, t)) {
                const uint32_t idf = ids[k];
                if ((idf >> kIdClassShift) == 0u || alpha_candidate<kAlphaTex>(sv, idf, make_f3(s.x, s.y, s.z), s.w, o, d, t)) {
                    const uint32_t id = idf & kIdMask;
                    if (t < best || (t == best && best_id != kMissId && id < best_id)) { best = t; best_id = id; }
                }
            }
        }
        if (sp == 0) break;
        sp -= stride;
        node = stack_decode(stack[sp]);
    }
    // t < tmax is required by intersect_sphere's contract: best starts at tmax and only shrinks
    t_out = best; id_out = best_id;
    if (kCount && visits) { visits[0] += n_nodes_visited; visits[1] += n_spheres_tested; }
}

// closest_hit through whichever view of the tree the scene offers: the LDS copy (binary records) when kLds, else the 4-wide view
// when the scene has one (SceneView::wide), else the binary records in global memory
template <bool kLds, typename StackT, bool kAlphaTex = false>
__device__ __forceinline__ void closest_hit_any(const SceneView& sv, const float4* __restrict__ nodes, const float4* __restrict__ sph,
                                                const uint32_t* __restrict__ ids, f3 o, f3 d, float tmin, float tmax, StackT* stack, uint32_t stride,
                                                float& t_out, uint32_t& id_out, uint32_t* visits = nullptr)
{
    // visits (global-memory scenes): this lane's running {node visits, sphere tests}, the scene term of SURVEY 8(d)'s byte accounting
    if (!kLds && visits) {
        if (sv.wide) closest_hit<StackT, true, true, kAlphaTex>(sv, sv.wide, sph, ids, sv.n, o, d, tmin, tmax, stack, stride, t_out, id_out, visits, sv.descent_cap);
        else closest_hit<StackT, true, false, kAlphaTex>(sv, nodes, sph, ids, sv.n, o, d, tmin, tmax, stack, stride, t_out, id_out, visits, sv.descent_cap);
    } else if (!kLds && sv.wide) {
        closest_hit<StackT, false, true, kAlphaTex>(sv, sv.wide, sph, ids, sv.n, o, d, tmin, tmax, stack, stride, t_out, id_out, nullptr, sv.descent_cap);
    } else {
        closest_hit<StackT, false, false, kAlphaTex>(sv, nodes, sph, ids, sv.n, o, d, tmin, tmax, stack, stride, t_out, id_out, nullptr, sv.descent_cap);
    }
}

// adds a workgroup's {node visits, sphere tests} to the lane's running totals (two atomics per workgroup)
__device__ __forceinline__ void flush_visit_counters(unsigned long long* totals, const uint32_t v[2])
{
    block_atomic_add(totals + 6, v[0]);
    block_atomic_add(totals + 7, v[1]);
}

// ------------------------------------------------------------------------------------------------ primary beams
// Camera rays of one 8x8-pixel block (= one wave64 of the primary pass) share their origin and span a thin pyramid.  One
// lane per block walks the BVH with that pyramid (four planes through the camera position, a pixel wider than the block on
// every side: half a pixel for any jitter in [-0.5, 0.5], half a pixel of slack) and lists the spheres whose padded leaf boxes it meets -- at most kBeamListCap; the primary pass then
// tests exactly those spheres for all 64 rays with wave-uniform control flow instead of 64 divergent stack traversals
// (DESIGN.md "Primary beams").  The list is a superset of every sphere any ray of the block can hit: a ray inside the
// pyramid that passes a leaf's padded box (the per-ray slab test's precondition for testing the sphere) means that box
// meets the pyramid; rounding in the plane tests is covered by the half-pixel widening plus an explicit relative margin.
// Closest hit over a superset with the same intersect_sphere and the same tie rule = the per-ray traversal's answer, bit
// for bit.  Record = 16 dwords: { count, original sphere ids[15] }; count > kBeamListCap = overflow, the wave traverses.
constexpr uint32_t kBeamListCap = 15;
constexpr uint32_t kBeamRecord = 16;

struct Beam {
    f3 o;
    f3 n[4];  // inward unit normals of the four side planes (zero vector = plane that culls nothing)
    float slack;  // every plane is moved outwards by this distance: the lists then hold for every camera position within `slack` of o
                  // (same orientation): a point x of the pyramid with apex o' has n.(x - o) = n.(x - o') + n.(o' - o) >= -|o' - o|
};

__device__ __forceinline__ f3 beam_plane(f3 a, f3 b, f3 inside)
{
    f3 n = cross(a, b);
    if (dot(n, inside) < 0.0f) n = -n;
    const float l2 = dot(n, n);
    if (!(l2 > 0.0f) || !is_finite(l2)) return make_f3(0.f, 0.f, 0.f);
    return n * __builtin_amdgcn_rsqf(l2);
}

__device__ __forceinline__ Beam make_beam(const CameraParams& cam, uint32_t px, uint32_t py, float slack, float margin_px = 0.0f)
{
    // NDC of the block's outline: its pixel centres lie in [px, px + 8] for every jitter in [-0.5, 0.5] (the host checks the
    // jitter), widened by half a pixel each way -- the lists serve every frame of a resting view
    // (margin_px more on every side: the lists then hold for every orientation whose rays leave the image within that many pixels of where
    // this one's do -- a camera that turns; pt_api.hip beam_cache_lookup bounds the displacement)
    const float xa = ((float)px - 0.5f - margin_px) * cam.InvW, xb = ((float)px + 8.5f + margin_px) * cam.InvW;
    const float ya = ((float)py - 0.5f - margin_px) * cam.InvH, yb = ((float)py + 8.5f + margin_px) * cam.InvH;
    const float nxa = pt_fma(xa, 2.0f, -1.0f), nxb = pt_fma(xb, 2.0f, -1.0f), nya = pt_fma(ya, -2.0f, 1.0f), nyb = pt_fma(yb, -2.0f, 1.0f);
    const f3 c00 = mad(nya, cam.Up, cam.Right * nxa) + cam.Forward, c10 = mad(nya, cam.Up, cam.Right * nxb) + cam.Forward;
    const f3 c11 = mad(nyb, cam.Up, cam.Right * nxb) + cam.Forward, c01 = mad(nyb, cam.Up, cam.Right * nxa) + cam.Forward;
    const f3 mid = (c00 + c11) + (c10 + c01);
    Beam b;
    b.o = cam.Position;
    b.slack = slack;
    b.n[0] = beam_plane(c00, c10, mid);
    b.n[1] = beam_plane(c10, c11, mid);
    b.n[2] = beam_plane(c11, c01, mid);
    b.n[3] = beam_plane(c01, c00, mid);
    return b;
}

// false = the box lies outside one of the planes for certain (NaNs compare false everywhere: never culled)
__device__ __forceinline__ bool beam_meets_box(const Beam& b, f3 lo, f3 hi)
{
    const f3 l = lo - b.o, h = hi - b.o;
    const float mag = pt_max(__builtin_fabsf(l.x), __builtin_fabsf(h.x)) + pt_max(__builtin_fabsf(l.y), __builtin_fabsf(h.y)) + pt_max(__builtin_fabsf(l.z), __builtin_fabsf(h.z));
    const float margin = -4e-6f * (mag + b.slack) - b.slack;
    bool meets = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const f3 n = b.n[k];
        // the box corner farthest along the inward normal
        const float d = pt_max(n.x * l.x, n.x * h.x) + pt_max(n.y * l.y, n.y * h.y) + pt_max(n.z * l.z, n.z * h.z);
        if (d < margin) meets = false;
    }
    return meets;
}

// leaf: the sphere that encloses the padded leaf box's inscribed sphere (centre = box centre, radius = largest half extent
// >= r + padding) against the planes
__device__ __forceinline__ bool beam_meets_leaf(const Beam& b, f3 lo, f3 hi)
{
    const f3 c = (lo + hi) * 0.5f - b.o;
    const f3 e = (hi - lo) * 0.5f;
    const float r = pt_max(e.x, pt_max(e.y, e.z));
    const float margin = -(r + b.slack + 4e-6f * (__builtin_fabsf(c.x) + __builtin_fabsf(c.y) + __builtin_fabsf(c.z) + r + b.slack));
    bool meets = true;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (dot(b.n[k], c) < margin) meets = false;
    return meets;
}

// One lane's work for one 8x8 block: walk the tree with the block's pyramid and write the block's record.
template <typename StackT>
__device__ __forceinline__ void beam_walk_block(const SceneView& sv, const float4* nodes, const uint32_t* ids, StackT* stack, uint32_t stride, const PixelMap& pm,
                                                const CameraParams& cam, float slack, float margin_px, uint32_t tile, uint32_t* __restrict__ lists)
{
    uint32_t* rec = lists + (size_t)tile * kBeamRecord;
    const PixelRef pr = slot_to_pixel(pm, tile << 6);  // the block's first pixel
    uint32_t count = 0;
    if (pr.valid) {
        const Beam b = make_beam(cam, pr.px, pr.py, slack, margin_px);
        if (sv.n == 1) {
            rec[1] = ids[0];
            count = 1;
        } else {
            int node = 0;
            uint32_t sp = 0;
            for (;;) {
                if (node >= 0) {
                    const float4 n0 = nodes[node * 4 + 0], n1 = nodes[node * 4 + 1], n2 = nodes[node * 4 + 2], n3 = nodes[node * 4 + 3];
                    const f3 lo0 = make_f3(n0.x, n0.y, n0.z), hi0 = make_f3(n0.w, n1.x, n1.y), lo1 = make_f3(n1.z, n1.w, n2.x), hi1 = make_f3(n2.y, n2.z, n2.w);
                    const int c0 = __builtin_bit_cast(int, n3.x), c1 = __builtin_bit_cast(int, n3.y);
                    const bool h0 = c0 >= 0 ? beam_meets_box(b, lo0, hi0) : (beam_meets_box(b, lo0, hi0) && beam_meets_leaf(b, lo0, hi0));
                    const bool h1 = c1 >= 0 ? beam_meets_box(b, lo1, hi1) : (beam_meets_box(b, lo1, hi1) && beam_meets_leaf(b, lo1, hi1));
                    if (h0 && h1) { stack[sp] = stack_encode<StackT>(c1); sp += stride; node = c0; continue; }
                    if (h0) { node = c0; continue; }
                    if (h1) { node = c1; continue; }
                } else {
                    if (count < kBeamListCap) rec[1 + count] = ids[~(uint32_t)node];
                    if (++count > kBeamListCap) break;  // overflow: the wave will traverse per ray
                }
                if (sp == 0) break;
                sp -= stride;
                node = stack_decode(stack[sp]);
            }
        }
    }
    rec[0] = count;
}

// kLds: the workgroup stages the tree into LDS first (LDS-resident scenes).  (Measured: 55 vs 58 us at 1080p -- a lane's walk is ~80 visits
// of ~150 instructions, bound by instruction issue of lone waves rather than by where the nodes live; the build stays off the frames'
// critical path: on a side stream for a resting view, behind a frame on its lane for a moving camera.  Sixteen lanes per block, each walking
// one subtree four levels down, were measured: the same 56 us beside the frames -- the build's time is its share of a busy chip, not its chain.)
template <bool kLds, typename StackT>
__global__ __launch_bounds__(256) void beam_kernel(SceneView sv, PixelMap pm, FrameParams fp, float slack, uint32_t* __restrict__ lists)
{
    extern __shared__ float4 smem[];
    const uint32_t stride = blockDim.x;
    const uint32_t n_blocks = pm.n_slots >> 6;
    const float4* nodes = sv.nodes;
    const uint32_t* ids = sv.sorted_id;
    StackT* stack;
    if (kLds) {
        stage_scene(sv, smem);
        nodes = smem;
        ids = reinterpret_cast<const uint32_t*>(smem + sv.n_nodes * 4u + sv.n);
        stack = reinterpret_cast<StackT*>(reinterpret_cast<char*>(smem) + scene_lds_bytes(sv.n_nodes, sv.n));
    } else {
        stack = reinterpret_cast<StackT*>(smem);
    }
    stack += threadIdx.x;
    for (uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x; tile < n_blocks; tile += gridDim.x * blockDim.x)
        beam_walk_block<StackT>(sv, nodes, ids, stack, stride, pm, fp.cam, slack, 0.0f, tile, lists);
}

// Closest hit of a primary ray over its block's candidate list (wave-uniform loop; sphere records come through the scalar
// cache).  Same intersect_sphere, same tie rule as closest_hit's leaves.
template <bool kAlphaTex = false>
__device__ __forceinline__ void closest_hit_list(const SceneView& sv, const uint32_t* __restrict__ rec, uint32_t count, f3 o, f3 d, float tmin, float tmax,
                                                 float& t_out, uint32_t& id_out)
{
    float best = tmax;
    uint32_t best_id = kMissId;
    for (uint32_t j = 0; j < count; j++) {
        const uint32_t idf = __builtin_amdgcn_readfirstlane(rec[1 + j]);  // (the list carries the leaf ids with their alpha class)
        const uint32_t id = idf & kIdMask;
        if ((idf >> kIdClassShift) == kAlphaInvisible) continue;  // wave-uniform
        const float4 s = sv.sph[id];
        float t;
        if (intersect_sphere(o, d, tmin, kInf, make_f3(s.x, s.y, s.z), s.w, t)) {
            if ((idf >> kIdClassShift) == 0u || alpha_candidate<kAlphaTex>(sv, idf, make_f3(s.x, s.y, s.z), s.w, o, d, t))
                if (t < best || (t == best && best_id != kMissId && id < best_id)) { best = t; best_id = id; }
        }
    }
    t_out = best; id_out = best_id;
}

// ------------------------------------------------------------------------------------------------ shade
__device__ __forceinline__ f3 load3(const float4& v) { return make_f3(v.x, v.y, v.z); }

// Per-path state carried between kernels in the ray queue (48 B) -- see RayQueue in pt_device.h.
struct PathState {
    f3 o, d, T;
    uint32_t slot, rng, bounce, sample;
    bool dirty;  // scratch.sample_rad[slot] holds this sample's radiance so far
    bool via_t;  // the sample left the primary surface through the transmission lobe (direct illumination does not cover it)
};

// One iteration of the bounce-loop body (Raytracing.hlsl:213-364) for a path whose ray (ps.o, ps.d) has been traced to
// (t, id); on sample end it accumulates into the pixel, and either finishes the pixel or regenerates the next sample
// from the cached primary hit and keeps going.  Returns true when ps holds a new ray that must be traced.
// What a hit needs for shading: geometry frame, the material after EvaluateMaterial (textures when kTex), BSDFSample.
struct HitMaterial {
    HitFrame hf;
    f3 emission, Ns;
    Bsdf bsdf;
};

template <bool kTex>
__device__ __forceinline__ HitMaterial hit_material(const SceneView& sv, uint32_t id, f3 o, f3 d, float t, bool primary)
{
    HitMaterial r;
    const float4 sp = sv.sph[id];
    const float4 m0 = sv.mats[id * 4 + 0], m1 = sv.mats[id * 4 + 1], m2 = sv.mats[id * 4 + 2], m3 = sv.mats[id * 4 + 3];
    r.hf = hit_frame(o, d, t, load3(sp), sp.w);
    f3 base = load3(m0), emissive_color = make_f3(m1.y, m1.z, m1.w);
    float metallic = m2.x, roughness = m2.y, transmission_m = m2.w;
    f3 Ns = r.hf.front ? r.hf.N : -r.hf.N;  // HitInfo.hlsli:60-64
    // (bit 31 of the device copy's AlphaMode word = "this sphere has texture maps", set by pt_set_textures: an untextured sphere in a textured
    // scene -- almost every hit of the demo -- costs no look-up of its map table, which would sit on the dependent chain of every bounce)
    if (kTex && (as_uint(m3.x) & kMaterialHasMaps) != 0u) {
        const uint4* mp = reinterpret_cast<const uint4*>(sv.tex_maps + (size_t)id * 8u);
        const uint4 ma = mp[0], mb = mp[1];
        {
            const uint32_t maps[kMapCount] = { ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z };
            float4 q;
            f3 n_mesh;
            const f2 uv = hit_uv(sv, id, r.hf.N, q, n_mesh);
            const f3 t_mesh = sphere_tangent(n_mesh);
            f3 T = quat_rotate(q.x, q.y, q.z, q.w, make_f3(t_mesh.x, t_mesh.y, -t_mesh.z));
            if (!r.hf.front) T = -T;  // HitInfo::GetFrontTangent
            const MaterialEval me = evaluate_material(sv.tex, maps, uv, base, m1.x, emissive_color, metallic, roughness, transmission_m, Ns, T);
            base = me.BaseColor; emissive_color = me.EmissiveColor; metallic = me.Metallic; roughness = me.Roughness;
            transmission_m = me.Transmission; Ns = me.Ns;
        }
    }
    r.emission = emissive_color * m1.x;  // Material::GetEmission
    r.Ns = Ns;
    // the primary hit mirrors the G-buffer round trip: Transmission = Metallic < 1 ? Transmission : 0 (Raytracing.hlsl:148)
    const float transmission = (primary && !(metallic < 1.0f)) ? 0.0f : transmission_m;
    // m3.z / m3.w: dielectric F0 and 1/IOR, precomputed per material by pt_set_scene (padding words of PtMaterial)
    r.bsdf = bsdf_init_pre(base, metallic, roughness, m2.z, m3.w, m3.z, transmission, r.hf.front);
    return r;
}

// kMulti = false specialises for SamplesPerPixel == 1: no radiance accumulator, no primary-hit cache, no sample
// regeneration (and with it no camera parameters live across the bounce loop).
// kTex = true adds EvaluateMaterial's texture branches + normal mapping (row N1; csrc/pt_texture.h).  It is a template
// parameter of every kernel that shades, chosen per launch from SceneView::tex_maps, so the kernels of the untextured hot
// path carry none of it (the textured fused kernels need 122-128 VGPRs against 101-107).
// Row N4: the direct-illumination estimate of a primary surface (csrc/pt_light.h; oracle render_pixel): ONE emissive sphere chosen
// uniformly, a direction uniformly inside the cone it subtends, a shadow ray through `trace` (the ordinary closest-hit query:
// the emitter must be the first thing it meets), DI = Le * (f_diffuse + f_specular) cos * n_lights / pdf with Le evaluated at the
// point the shadow ray reaches (EvaluateMaterial: an emissive map modulates it).  Own per-pixel RNG stream.
template <bool kTex, typename TraceFn>
__device__ __forceinline__ f3 di_estimate(const SceneView& sv, const FrameParams& fp, uint32_t px, uint32_t py, uint32_t id, f3 d, const HitMaterial& hm,
                                          TraceFn&& trace, uint32_t& rays)
{
    f3 est = make_f3(0.f, 0.f, 0.f);
    uint32_t rng = rng_init(px, py, fp.frame_index ^ kDiRngSalt);
    const float u0 = rng_float(rng), u1 = rng_float(rng), u2 = rng_float(rng);
    const uint32_t light = sv.lights[pick_light(u0, sv.n_lights)];
    const float4 ls = sv.sph[light];
    const LightSample s = sample_sphere_cone(hm.hf.P, load3(ls), ls.w, u1, u2);
    const Surf surf = surf_init(hm.hf.front, hm.hf.N, hm.Ns);
    if (light != id && s.valid && dot(surf.FrontNg, s.L) > 0.0f) {
        const f3 V = -d;
        float w[3];
        lobe_weights(hm.bsdf, surf, V, w);
        const f3 f = bsdf_eval_reflective(hm.bsdf, surf, s.L, V, w);
        // No shadow ray for a contribution that cannot matter: the estimate's upper bound with the emitter's untextured radiance
        // (maps modulate it downwards) is below kDiNegligible -- a mirror-like primary surface seen off its specular direction,
        // i.e. most of the demo's ground.  (Bias below 1e-7 of unit radiance per pixel; spec of this row, not of the reference.)
        const float4 lm = sv.mats[light * 4 + 1];  // {EmissiveStrength, EmissiveColor}
        const float k = s.inv_pdf * (float)sv.n_lights;
        const float bound = pt_max(f.x * lm.y, pt_max(f.y * lm.z, f.z * lm.w)) * (lm.x * k);
        if (bound > kDiNegligible) {
            float t2;
            uint32_t id2;
            const f3 so = spawn_origin(hm.hf.P, hm.hf.N, hm.hf.offset, s.L);
            trace(so, s.L, t2, id2);
            rays++;
            if (id2 == light) {
                const f3 le = hit_material<kTex>(sv, light, so, s.L, t2, false).emission;
                est = (le * f) * k;
            }
        }
    }
    if (!(est.x > 0.0f || est.y > 0.0f || est.z > 0.0f) || !is_finite(est.x) || !is_finite(est.y) || !is_finite(est.z))
        est = make_f3(0.f, 0.f, 0.f);  // NaN / inf / negative estimates count as no light
    return est;
}

struct NoTrace {
    __device__ __forceinline__ void operator()(f3, f3, float&, uint32_t&) const {}
};

// kDI = true (primary passes of the fused schedule): the direct-illumination estimate is made HERE, at the first shading of the
// primary surface, sharing its material / BSDF evaluation; `trace` casts the shadow ray, `di_rays` counts it.  With kDI = false
// and fp.di_enabled the estimate is read from scratch.di (written by a kDI pass or, in the split schedule, by di_kernel).
// kCacheMode (kMulti, untextured kernels; Scratch::primary_cache): 1 = the primary pass stores what the first shading of the primary
// surface computed (plus where the pixel is written); 2 = the looping pass restarts a pixel's next sample from that record instead of recomputing the primary ray, the
// hit frame and the lobe weights (~250 of the ~1100 instructions of a full shading step, once per sample: all samples of a pixel share
// their primary hit, Raytracing.hlsl:193-198).  The cached values are the ones the full path computes, bit for bit.
// kMerge (the looping pass at spp > 1): ONE pass through the surface-shading code per call.  The plain form shades the traced hit,
// and when that ends the sample loops back to shade the regenerated primary hit -- two passes through the most expensive code of the
// kernel, each for a fraction of the wave's lanes (92 % of C3's bounce rays miss: the first pass runs for the few lanes that hit).
// Merged, a lane whose ray missed finishes its sample first and then shades its regenerated primary hit TOGETHER with the lanes that
// hit; a lane whose sample ends in the surface code returns kShadePending and starts its next call there (no ray to trace in between).
// Per-lane order of operations, RNG draws and arithmetic are unchanged.  `pending` in: the previous call returned kShadePending.
enum : int { kShadeDone = 0, kShadeRay = 1, kShadePending = 2 };

template <bool kMulti, bool kTex = false, bool kDI = false, int kCacheMode = 0, bool kMerge = false, typename TraceFn = NoTrace>
__device__ __forceinline__ int shade_step_ex(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const Scratch& scratch,
                                             float4* __restrict__ out, PathState& ps, float t, uint32_t id, bool pending, TraceFn&& trace = NoTrace(), uint32_t* di_rays = nullptr)
{
    static_assert(!kMerge || (kMulti && !kDI), "the merged form is the looping pass of spp > 1 frames");
    const uint32_t slot = ps.slot;
    f3 di_val = make_f3(0.f, 0.f, 0.f);
    bool di_have = false;  // di_val is this pixel's estimate, made in this call
    bool regen = false;    // kCacheMode 2: the surface being shaded is a regenerated primary hit described by the pixel's cached record
    float4 rc0 = make_float4(0.f, 0.f, 0.f, 0.f), rc1 = rc0;
    // sampleRadiance so far is (dirty ? scratch : 0); srad_loaded says whether `srad` holds it
    f3 srad = make_f3(0.f, 0.f, 0.f);
    bool srad_loaded = false, srad_changed = false;
    bool end_sample = false;
    f3 L = make_f3(0.f, 0.f, 0.f);
    HitFrame hf;

    // the ray missed: sampleRadiance += T * environment (:242-259); a primary miss writes the pixel and returns true
    auto shade_miss = [&]() -> bool {
        f3 env;
        if (kTex && sv.env_tex != kNoTexture)
            env = sv.env_cube ? environment_cube(sv.tex + sv.env_tex, sv.env_xf, ps.d) : environment_texture(sv.tex[sv.env_tex], sv.env_xf, ps.d);
        else env = environment_color(sv.env[0], sv.env[1], sv.env[2], sv.env[3], ps.d);
        if (!kMerge && ps.bounce == 0) {  // primary miss: pixel = environment (GBufferGeneration.hlsl:223-227; Raytracing.hlsl:249-252); the looping pass only sees paths with a primary hit
            out[slot_to_pixel(pm, slot).out_index] = make_float4(env.x, env.y, env.z, 1.0f);
            return true;
        }
        if (ps.dirty) { const float4 s = scratch.sample_rad[slot]; srad = load3(s); }
        srad_loaded = true;
        srad = srad + ps.T * env;  // :254
        end_sample = true;
        return false;
    };

    // the ray hit sphere `id` at t (or: the regenerated primary hit): material, emission, lobe choice, sample, pdf, f, throughput,
    // Russian roulette, cut-off (:293-364).  Sets end_sample, or leaves L / hf for the spawn.
    auto shade_surface = [&]() {
        HitMaterial hm;
        if (kCacheMode == 2 && regen) {
            // the primary hit of this pixel again: frame and weights from the record (rc0 = {N, offset}, rc1 = {weights, id}), the
            // material as hit_material forms it (bounce 0: Transmission = Metallic < 1 ? Transmission : 0, Raytracing.hlsl:148)
            const float4 sp = sv.sph[id];
            const float4 m0 = sv.mats[id * 4 + 0], m1 = sv.mats[id * 4 + 1], m2 = sv.mats[id * 4 + 2], m3 = sv.mats[id * 4 + 3];
            hm.hf.N = make_f3(rc0.x, rc0.y, rc0.z);
            hm.hf.P = mad(sp.w, hm.hf.N, load3(sp));
            hm.hf.offset = rc0.w;
            hm.hf.front = dot(hm.hf.N, ps.d) < 0.0f;
            hm.Ns = hm.hf.front ? hm.hf.N : -hm.hf.N;
            hm.emission = make_f3(m1.y, m1.z, m1.w) * m1.x;
            hm.bsdf = bsdf_init_pre(load3(m0), m2.x, m2.y, m2.z, m3.w, m3.z, !(m2.x < 1.0f) ? 0.0f : m2.w, hm.hf.front);
        } else {
            hm = hit_material<kTex>(sv, id, ps.o, ps.d, t, ps.bounce == 0);
        }
        hf = hm.hf;
        f3 emission = hm.emission;
        const f3 Ns = hm.Ns;
        const Bsdf& bsdf = hm.bsdf;
        // Sphere-light direct illumination (row N4) covers what the reflective lobes of the primary surface receive from the
        // emitters, so the emission of a first-bounce hit reached through them is dropped (Raytracing.hlsl:302).  The flag
        // is "the pixel has a primary surface", not the reference's any(DI > 0): with a one-sample estimator DI = 0 is an
        // ordinary sample value and conditioning on it would bias the frame upward; and a sample that left through the
        // transmission lobe keeps its emission, because DI evaluates the reflective lobes only.
        if (fp.di_enabled && ps.bounce == 1 && !ps.via_t) emission = make_f3(0.f, 0.f, 0.f);
        if (kDI && fp.di_enabled && ps.bounce == 0 && (!kMulti || ps.sample == 0) && !di_have) {
            const PixelRef dpr = slot_to_pixel(pm, slot);
            di_val = di_estimate<kTex>(sv, fp, dpr.px, dpr.py, id, ps.d, hm, trace, *di_rays);
            di_have = true;
        }
        const bool t_finite = is_finite(ps.T.x) && is_finite(ps.T.y) && is_finite(ps.T.z);
        if (emission.x != 0.0f || emission.y != 0.0f || emission.z != 0.0f || !t_finite) {
            if (ps.dirty) { const float4 s = scratch.sample_rad[slot]; srad = load3(s); }
            srad_loaded = true;
            srad = srad + ps.T * emission;  // :320
            srad_changed = true;
        }
        const bool last = ps.bounce == fp.bounces;
        if (last && (!kMulti || ps.sample + 1 == fp.spp)) {
            end_sample = true;  // the sample drawn on the final iteration is never used and no later sample reads the RNG
            return;
        }
        const Surf surf = surf_init(hf.front, hf.N, Ns);
        const f3 V = -ps.d;
        float w[3];
        if (kCacheMode == 2 && regen) {
            w[0] = rc1.x; w[1] = rc1.y; w[2] = rc1.z;
        } else {
            lobe_weights(bsdf, surf, V, w);
            if (kCacheMode == 1 && scratch.primary_cache && ps.bounce == 0 && ps.sample == 0) {
                float4* rec = scratch.primary_cache + (size_t)slot * 3u;
                rec[0] = make_float4(hf.N.x, hf.N.y, hf.N.z, hf.offset);
                rec[1] = make_float4(w[0], w[1], w[2], as_float(id));
                rec[2] = make_float4(ps.d.x, ps.d.y, ps.d.z, as_float(slot_to_pixel(pm, slot).out_index));  // (where the pixel is written: the looping pass needs no slot -> pixel mapping)
            }
        }
        float rnd[4];
        rnd[0] = rng_float(ps.rng); rnd[1] = rng_float(ps.rng); rnd[2] = rng_float(ps.rng); rnd[3] = rng_float(ps.rng);  // :330
        int lobe;
        if (!bsdf_sample(bsdf, surf, V, w, rnd, L, lobe)) { end_sample = true; return; }
        float pdf;
        f3 f;
        if (!bsdf_pdf_eval(bsdf, surf, L, V, w, lobe, pdf, f)) { end_sample = true; return; }  // pdf == 0 (:336-339)
        if (f.x == 0.0f && f.y == 0.0f && f.z == 0.0f) { end_sample = true; return; }
        { const float inv_pdf = pt_rcp(pdf); ps.T = ps.T * (f * inv_pdf); }  // :346
        if (ps.bounce == 0) ps.via_t = lobe == kLobeTransmission;
        if (fp.rr_enabled && ps.bounce > 3) {                     // :348-356
            const float p = pt_max(ps.T.x, pt_max(ps.T.y, ps.T.z));
            if (rng_float(ps.rng) >= p) end_sample = true;
            else ps.T = ps.T * pt_rcp(p);
        }
        if (!end_sample && luminance(ps.T) <= fp.throughput_threshold) end_sample = true;  // :361
        if (last) end_sample = true;
    };

    // spawn the next ray (Raytracing.hlsl:219-224)
    auto spawn = [&]() {
        ps.o = spawn_origin(hf.P, hf.N, hf.offset, L);
        ps.d = L;
        ps.bounce++;
        if (srad_changed) { scratch.sample_rad[slot] = make_float4(srad.x, srad.y, srad.z, 0.f); ps.dirty = true; }
        if (kDI && di_have) scratch.di[slot] = make_float4(di_val.x, di_val.y, di_val.z, 0.f);  // the pass that finishes the pixel adds it
    };

    // end of sample: radiance += sampleRadiance (:373); true = that was the pixel's last sample, the pixel has been written (:378-385)
    auto finish_sample = [&]() -> bool {
        if (!srad_loaded) {
            if (ps.dirty) { const float4 s = scratch.sample_rad[slot]; srad = load3(s); }
        }
        f3 acc = make_f3(0.f, 0.f, 0.f);
        if (kMulti && ps.sample > 0) { const float4 r = scratch.radiance[slot]; acc = load3(r); }
        const f3 total = acc + srad;
        ps.sample++;
        if (!kMulti || ps.sample == fp.spp) {
            f3 res = make_f3(0.f, 0.f, 0.f);
            if (is_finite(total.x) && is_finite(total.y) && is_finite(total.z)) {
                res = total * fp.inv_spp;
            }
            if (fp.di_enabled) {  // radiance += DI (:381)
                if (kDI && di_have) res = res + di_val;
                else { const float4 di = scratch.di[slot]; res = res + load3(di); }
            }
            const uint32_t out_index = kCacheMode == 2 ? as_uint(scratch.primary_cache[(size_t)slot * 3u + 2u].w) : slot_to_pixel(pm, slot).out_index;
            out[out_index] = make_float4(res.x, res.y, res.z, 1.0f);
            return true;
        }
        scratch.radiance[slot] = make_float4(total.x, total.y, total.z, 0.f);
        return false;
    };

    // regenerate: every sample restarts from the same primary ray / primary hit (:193-198)
    auto regenerate = [&]() {
        if (kCacheMode == 2) {  // (the host hands kernels of this mode the records: render_common)
            const float4* rec = scratch.primary_cache + (size_t)slot * 3u;
            rc0 = rec[0]; rc1 = rec[1];
            const float4 rc2 = rec[2];
            ps.o = fp.cam.Position;
            ps.d = make_f3(rc2.x, rc2.y, rc2.z);
            id = as_uint(rc1.w);
            regen = true;
        } else {
            const PixelRef pr = slot_to_pixel(pm, slot);
            float tmin, tmax;
            primary_ray(fp.cam, pr.px, pr.py, ps.o, ps.d, tmin, tmax);
            const uint2 ph = scratch.primary_hit[slot];
            t = as_float(ph.x);
            id = ph.y;
        }
        ps.T = make_f3(1.f, 1.f, 1.f);
        ps.bounce = 0;
        ps.dirty = false;
        ps.via_t = false;
        srad = make_f3(0.f, 0.f, 0.f);
        srad_loaded = false; srad_changed = false;
        end_sample = false;
    };

    if (kMerge) {
        if (pending) {
            regenerate();
        } else if (id == kMissId) {
            if (shade_miss()) return kShadeDone;
            if (finish_sample()) return kShadeDone;
            regenerate();
        }
        shade_surface();  // the traced hit, or the regenerated primary hit of a lane whose ray missed / whose last call ended its sample
        if (!end_sample) { spawn(); return kShadeRay; }
        return finish_sample() ? kShadeDone : kShadePending;
    }
    for (;;) {
        if (id == kMissId) { if (shade_miss()) return kShadeDone; }
        else shade_surface();
        if (!end_sample) { spawn(); return kShadeRay; }
        if (finish_sample()) return kShadeDone;
        regenerate();
    }
}

// One iteration of the bounce-loop body for a path whose ray (ps.o, ps.d) has been traced to (t, id), as the kernels outside the merged
// looping pass use it: true = ps holds a new ray that must be traced, false = the pixel is finished.
template <bool kMulti, bool kTex = false, bool kDI = false, int kCacheMode = 0, typename TraceFn = NoTrace>
__device__ __forceinline__ bool shade_step(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const Scratch& scratch,
                                           float4* __restrict__ out, PathState& ps, float t, uint32_t id, TraceFn&& trace = NoTrace(), uint32_t* di_rays = nullptr)
{
    return shade_step_ex<kMulti, kTex, kDI, kCacheMode, false>(sv, pm, fp, scratch, out, ps, t, id, false, trace, di_rays) == kShadeRay;
}

__device__ __forceinline__ PathState load_path(const RayQueue& q, uint32_t i)
{
    const float4 a = q.q0[i], b = q.q1[i], c = q.q2[i];
    PathState ps;
    ps.o = load3(a); ps.d = load3(b); ps.T = load3(c);
    ps.slot = as_uint(a.w); ps.rng = as_uint(b.w);
    const uint32_t flags = as_uint(c.w);
    ps.bounce = flags & kFlagBounceMask;
    ps.sample = (flags >> kFlagSampleShift) & kFlagSampleMask;
    ps.dirty = (flags & kFlagDirty) != 0;
    ps.via_t = (flags & kFlagViaTransmission) != 0;
    return ps;
}

__device__ __forceinline__ void store_path(const RayQueue& q, uint32_t j, const PathState& ps)
{
    const uint32_t flags = (ps.bounce & kFlagBounceMask) | ((ps.sample & kFlagSampleMask) << kFlagSampleShift) | (ps.dirty ? kFlagDirty : 0u)
                           | (ps.via_t ? kFlagViaTransmission : 0u);
    q.q0[j] = make_float4(ps.o.x, ps.o.y, ps.o.z, as_float(ps.slot));
    q.q1[j] = make_float4(ps.d.x, ps.d.y, ps.d.z, as_float(ps.rng));
    q.q2[j] = make_float4(ps.T.x, ps.T.y, ps.T.z, as_float(flags));
}

// static LDS of the kernels below (counters, the segment prefix table of the looping pass) on top of their dynamic LDS: the
// 64 KB default limit counts both, so the opt-in for more is taken this much earlier
constexpr uint32_t kStaticLdsMargin = (kMaxSegs + 64u) * 4u;

// Cold kernel arguments.  A by-value kernel argument is loaded into SGPRs at the kernel's entry and stays there: with ~120 dwords of
// arguments and 100 SGPRs, the camera and the pixel map -- needed once per batch, to generate the primary rays -- lived in VGPR lanes
// and came back through ~70 v_readlane per batch (VALU issue slots, in a VALU-bound kernel).  cold_arg re-reads such an argument from the
// kernarg segment where it is used (s_load, scalar cache): the laundered pointer keeps the loads inside the loop.
// The leading arguments of bounce_kernel, as the kernarg segment lays them out (in order, each at its natural alignment = a C struct):
struct BounceArgHead { SceneView sv; PixelMap pm; FrameParams fp; };
template <typename T>
__device__ __forceinline__ T cold_arg(uint32_t offset)
{
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    T v;
    __builtin_memcpy(&v, p + offset, sizeof(T));
    return v;
}

// ------------------------------------------------------------------------------------------------ fused bounce
// trace + shade in one kernel (DESIGN.md "Kernels"): a lane obtains a ray (kPrimary: generated from its pixel; else read
// from the input queue), traces it, and runs one shade_step.  kLoop = false: survivors are compacted into the output
// queue (wave64 ballot + prefix, one atomic per workgroup) -- one wavefront bounce per launch, 96 B of queue traffic
// per ray and no hit stream.  kLoop = true: the lane keeps alternating trace and shade_step until its pixel is done
// (the persistent "tail" form for small queues).
template <bool kLds, typename StackT, bool kPrimary, bool kLoop, bool kMulti, bool kTex, bool kInline2, bool kFuse, bool kDI>
// 4 waves/SIMD (<= 128 VGPRs): two 512-thread workgroups per CU with the BVH in LDS (the unconstrained build takes 134
// VGPRs for the primary variant and drops to 3 waves/SIMD: measured 171 -> 149 us for the two compacting passes at C2)
__global__ __launch_bounds__(kFusedThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void bounce_kernel(SceneView sv, PixelMap pm, FrameParams fp, RayQueue qin, RayQueue qout,
                                                               Scratch scratch, float4* __restrict__ out,
                                                               const uint32_t* __restrict__ count_in_ptr, uint32_t* __restrict__ count_out_ptr,
                                                               FrameCounters fc)
{
    // Bounces traced per lane before the survivors are compacted.  The primary pass of a 1-spp frame takes TWO (the primary
    // ray and, in registers, the first bounce ray of the ~49 % of pixels that hit something): the ~1.0 M bounce-1 rays of a
    // 1080p frame then never travel through HBM (96 MB per frame) and the most latency-exposed pass of the frame (a
    // separate queue-fed bounce-1 launch, 63 % of its wave time in s_waitcnt) disappears; the half-empty waves cost less
    // than that.  Measured on C2: 1 bounce 0.1165, 2 bounces 0.107, 3 bounces 0.116, 4 bounces 0.123 ms per frame (kInline2 is
    // the host's switch; it is on for every 1-spp frame of the fused schedule).
    constexpr uint32_t kIters = (kPrimary && !kLoop && !kMulti && kInline2) ? 2u : 1u;
    // The looping pass of a 1-spp frame is a chain of dependent bounces on a few thousand nearly empty waves that share their SIMDs with the
    // primary passes of the frames behind it.  VALU issue is arbitrated by priority, then age (MI355X_MICROARCH.md), and those primary waves are
    // usually the older ones: raised priority lets the chain run as if alone, for a handful of issue slots taken from the throughput-bound waves.
    if (kLoop && !kMulti && !kPrimary) __builtin_amdgcn_s_setprio(3);
    extern __shared__ float4 smem[];
    __shared__ uint32_t s_wave_count[kFusedThreads / 64];
    __shared__ uint32_t s_block_base;
    __shared__ uint32_t s_seg_count;                  // producer side of the segmented hand-over (FrameCounters::seg_counts)
    __shared__ uint32_t s_seg_next;                   // ... and the next 64-slot tile of this workgroup's batches to hand to a wave
    __shared__ uint32_t s_loop_next;                  // fused form: the next 64 entries of the workgroup's own segment
    __shared__ uint32_t s_seg_prefix[kMaxSegs + 1];   // consumer side: s_seg_prefix[b] = entries in segments < b
    if (kPrimary && blockIdx.x == 0) frame_counters_begin(fc, pm.n_slots);
    const bool seg_out = kPrimary && !kLoop && fc.seg_counts != nullptr;
    const bool seg_in = !kPrimary && kLoop && fc.seg_counts != nullptr;
    uint32_t count;
    if (seg_in) {
        // exclusive prefix sum of the segment sizes: thread t sums a run of consecutive segments, the runs are scanned per wave
        // (shuffles) and across waves (s_wave_count), then every thread writes the prefixes of its run
        const uint32_t per = (fc.n_segs + blockDim.x - 1u) / blockDim.x;
        const uint32_t first = threadIdx.x * per;
        uint32_t sum = 0;
        for (uint32_t j = 0; j < per; j++) if (first + j < fc.n_segs) sum += fc.seg_counts[first + j];
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if ((int)lane_id() >= off) incl += v; }
        if (lane_id() == 63u) s_wave_count[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += s_wave_count[w];
        uint32_t run = base + incl - sum;
        for (uint32_t j = 0; j < per; j++)
            if (first + j < fc.n_segs) { s_seg_prefix[first + j] = run; run += fc.seg_counts[first + j]; }
        if (threadIdx.x == blockDim.x - 1u) s_seg_prefix[fc.n_segs] = base + incl;
        __syncthreads();
        count = s_seg_prefix[fc.n_segs];
        if (blockIdx.x == 0 && threadIdx.x == 0) *const_cast<uint32_t*>(count_in_ptr) = count;  // the queue size, for the statistics and the next frame's grid estimate
    } else {
        count = kPrimary ? pm.n_slots : *count_in_ptr;
    }
    if (seg_out && threadIdx.x == 0) { s_seg_count = 0; s_seg_next = 0; s_loop_next = 0; }
    if (blockIdx.x * blockDim.x >= count) {
        if (seg_out && threadIdx.x == 0) fc.seg_counts[blockIdx.x] = 0;
        return;
    }
    if (seg_out && !kLds) __syncthreads();  // (with kLds the barrier of stage_scene orders the reset)
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
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t my_rays = 0;
    // A share of the NEXT primary-beam lists of a moving camera (FrameParams::beam_job, pt_api.hip beam_cache_lookup): the first wave of the
    // first workgroups walks 64 blocks' pyramids each, one lane per block, before it joins its workgroup's tiles -- an 8192-block share per
    // frame hides in the pass (its other waves take the tiles meanwhile), where a whole build as a launch of its own stalled a lane for 60 us
    // and put the three lanes out of step for several frames (tools/experiments/moving_trace.sh).
    if (kPrimary && !kLoop && wave == 0) {
        const BeamJob job = cold_arg<BeamJob>(offsetof(BounceArgHead, fp) + offsetof(FrameParams, beam_job));
        const uint32_t j = blockIdx.x * 64u + lane;
        if (j < job.n_blocks) {
            CameraParams cam_b = cold_arg<CameraParams>(offsetof(BounceArgHead, fp) + offsetof(FrameParams, cam));
            cam_b.Position = make_f3(job.centre[0], job.centre[1], job.centre[2]);
            cam_b.Right = make_f3(job.right[0], job.right[1], job.right[2]);  // (the orientation the lists are made for: this frame's, or one extrapolated from a turn)
            cam_b.Up = make_f3(job.up[0], job.up[1], job.up[2]);
            cam_b.Forward = make_f3(job.forward[0], job.forward[1], job.forward[2]);
            beam_walk_block<StackT>(sv, nodes, ids, stack, blockDim.x, cold_arg<PixelMap>(offsetof(BounceArgHead, pm)), cam_b, job.slack, job.margin_px, job.first_block + j, job.lists);
        }
    }
    // Work distribution: a static grid-stride over the batches -- except for the looping pass behind a segmented hand-over,
    // where every WAVE pulls its next 64 entries from a work cursor (one atomic per 64 paths; no barrier inside this loop in
    // the looping form): a segment lists one workgroup's tiles top to bottom of the image, and a static stride over such a
    // queue can hand a workgroup the same image region again and again (C3: 7.2 vs 6.5 ms for the looping pass).
    // 1 spp (kMulti = false): every wave's FIRST 64 entries are its own by position (no atomic); only what lies beyond one entry
    // per thread of the grid is handed out by the cursor.  The host sizes the grid to the queue, so the pass makes no cursor
    // atomics at all: 1200 waves hitting one address at launch cost 5-10 us EACH (tools/experiments/loopstamps.py: same-address
    // device-scope atomics from 8 XCDs serialise), before the first ray and again to learn that the queue is empty -- the
    // driver's 20-step C2 run went from 0.102-0.106 to 0.095 ms per frame.  At spp > 1 a wave draws ~30 long batches, the atomics
    // do not show, and the extra loop state made that kernel spill (C4 +3 %): it keeps the plain cursor.
    uint32_t* const cursor = const_cast<uint32_t*>(count_in_ptr) + fc.n_counts + 1u;  // the work cursor of this queue (zeroed when the counters are folded)
    for (uint32_t base = blockIdx.x * blockDim.x;; base += gridDim.x * blockDim.x) {
        uint32_t i;
        if (seg_in) {
            uint32_t b;
            const uint32_t static_end = kMulti ? 0u : gridDim.x * blockDim.x;
            if (!kMulti && base < static_end) {  // the first pass through this loop
                b = base + wave * 64u;
            } else {
                if (!kMulti && static_end >= count) break;
                uint32_t v = 0;
                if (lane == 0) v = atomicAdd(cursor, 64u);
                b = static_end + __builtin_amdgcn_readfirstlane(v);
            }
            if (b >= count) break;
            i = b + lane;
        } else if (seg_out) {
            // the workgroup's batches are the same strided set as below, but its waves take the 64-slot tiles of them from a
            // counter in LDS: a wave that drew sky tiles moves on instead of idling behind its neighbours
            uint32_t w = 0;
            if (lane == 0) w = atomicAdd(&s_seg_next, 1u);
            w = __builtin_amdgcn_readfirstlane(w);
            const uint32_t waves = blockDim.x >> 6;
            const uint32_t b = (blockIdx.x + (w / waves) * gridDim.x) * blockDim.x;
            if (b >= count) break;
            i = b + (w % waves) * 64u + lane;
        } else {
            if (base >= count) break;
            i = base + threadIdx.x;
        }
        bool emit = false;
        PathState ps;
        // primary beams: this wave's 64 slots are one 8x8-pixel block; its candidate list was made by beam_kernel
        const uint32_t* beam_rec = nullptr;
        uint32_t beam_count = ~0u;
        if (kPrimary && fp.beam_lists) {
            beam_rec = fp.beam_lists + (size_t)__builtin_amdgcn_readfirstlane(i >> 6) * kBeamRecord;
            beam_count = __builtin_amdgcn_readfirstlane(beam_rec[0]);
        }
        if (i < count) {
            bool live = true;
            float tmin = 0.0f, tmax = kInf;
            if (kPrimary) {
                const PixelMap pm_c = cold_arg<PixelMap>(offsetof(BounceArgHead, pm));
                const PixelRef pr = slot_to_pixel(pm_c, i);
                live = pr.valid;
                ps.slot = i; ps.bounce = 0; ps.sample = 0; ps.dirty = false; ps.via_t = false; ps.rng = 0;
                ps.T = make_f3(1.f, 1.f, 1.f);
                ps.o = make_f3(0.f, 0.f, 0.f); ps.d = make_f3(0.f, 0.f, 1.f);
                if (live) {
                    const CameraParams cam_c = cold_arg<CameraParams>(offsetof(BounceArgHead, fp) + offsetof(FrameParams, cam));
                    primary_ray(cam_c, pr.px, pr.py, ps.o, ps.d, tmin, tmax);
                    ps.rng = rng_init(pr.px, pr.py, fp.frame_index);
                } else if (pm_c.mode == 1) {
                    out[pr.out_index] = make_float4(0.f, 0.f, 0.f, 0.f);  // padding pixel of an edge tile
                }
            } else if (seg_in) {
                // dense index -> (segment, offset): the last segment whose prefix is <= i
                uint32_t lo = 0, hi = fc.n_segs;
                while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_seg_prefix[mid] <= i) lo = mid; else hi = mid; }
                ps = load_path(qin, lo * fc.seg_cap + (i - s_seg_prefix[lo]));
            } else {
                ps = load_path(qin, i);
            }
            constexpr bool kMerged = kLoop && kMulti && !kPrimary;  // (shade_step_ex: one pass through the surface code per iteration)
            if (kMerged) {
                constexpr int kCacheMode = kTex ? 0 : 2;
                bool pending = false;
                for (;;) {
                    float t = 0.0f;
                    uint32_t id = kMissId;
                    if (!pending) closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, ps.o, ps.d, 0.0f, kInf, stack, blockDim.x, t, id);
                    const int r = shade_step_ex<true, kTex, false, kCacheMode, true>(sv, pm, fp, scratch, out, ps, t, id, pending);
                    if (r == kShadeDone) break;
                    pending = r == kShadePending;
                    if (!pending) my_rays++;  // a ray spawned and traced inside this kernel
                }
            } else if (live) {
                bool primary_trace = kPrimary;
                uint32_t iter = 0;
                for (;;) {
                    float t;
                    uint32_t id;
                    if (kPrimary && primary_trace && fp.beam_lists && beam_count <= kBeamListCap) {
                        closest_hit_list<kTex>(sv, beam_rec, beam_count, ps.o, ps.d, tmin, tmax, t, id);
                        if (kMulti) scratch.primary_hit[i] = make_uint2(as_uint(t), id);
                    } else {
                        closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, ps.o, ps.d, tmin, tmax, stack, blockDim.x, t, id);
                        if (kMulti && kPrimary && primary_trace) scratch.primary_hit[i] = make_uint2(as_uint(t), id);
                    }
                    primary_trace = false;
                    // Scratch::primary_cache: the compacting primary pass of an spp > 1 frame writes the records, the looping pass restarts samples from them
                    constexpr int kCacheMode = (kMulti && !kTex) ? ((kPrimary && !kLoop) ? 1 : ((kLoop && !kPrimary) ? 2 : 0)) : 0;
                    if (kDI)  // row N4: the first shading of the primary surface also makes its direct-illumination estimate
                        emit = shade_step<kMulti, kTex, true, kCacheMode>(sv, pm, fp, scratch, out, ps, t, id,
                                                              [&](f3 so, f3 sd, float& t2, uint32_t& id2) { closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, so, sd, 0.0f, kInf, stack, blockDim.x, t2, id2); },
                                                              &my_rays);
                    else
                        emit = shade_step<kMulti, kTex, false, kCacheMode>(sv, pm, fp, scratch, out, ps, t, id);
                    if (!emit) break;
                    if (!kLoop && ++iter >= kIters) break;
                    my_rays++;  // a ray spawned and traced inside this kernel (queued rays are counted by counts[])
                    tmin = 0.0f; tmax = kInf;
                }
            }
        }
        if (seg_out) {
            // Segmented hand-over: the wave reserves room in the workgroup's own segment with ONE LDS atomic; no barrier -- the
            // waves of a workgroup no longer wait for its slowest wave after every batch (the barriers of the dense form below
            // cost 18 % of the primary pass at C2: 132 -> 109 us measured with the compaction compiled out).
            const unsigned long long mask = __ballot(emit);
            const uint32_t wave_n = __popcll(mask);
            uint32_t base = 0;
            if (lane == 0 && wave_n) base = atomicAdd(&s_seg_count, wave_n);
            base = __shfl(base, 0, 64);
            if (emit) store_path(qout, blockIdx.x * fc.seg_cap + base + __popcll(mask & ((1ull << lane) - 1ull)), ps);
        } else if (!kLoop) {
            // One atomic per WORKGROUP.  (One per wave -- no barriers, waves never wait for each other -- was measured and is
            // far worse: 0.116 -> 0.201 ms per C2 frame; 8x as many same-address device-scope atomics from 8 XCDs serialise.)
            const unsigned long long mask = __ballot(emit);
            const uint32_t wave_n = __popcll(mask);
            const uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave_count[wave] = wave_n;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t total = 0;
                const uint32_t n_waves = blockDim.x >> 6;
                for (uint32_t w = 0; w < n_waves; w++) { const uint32_t c = s_wave_count[w]; s_wave_count[w] = total; total += c; }
                s_block_base = total ? atomicAdd(count_out_ptr, total) : 0u;
            }
            __syncthreads();
            if (emit) store_path(qout, s_block_base + s_wave_count[wave] + prefix, ps);
            __syncthreads();
        }
    }
    uint32_t my_loop_rays = 0;
    if (seg_out) {
        __syncthreads();  // (also orders the queue records written above before the reads below, workgroup scope)
        const uint32_t n_own = s_seg_count;
        if (threadIdx.x == 0) {
            fc.seg_counts[blockIdx.x] = n_own;
            if (kFuse && n_own) atomicAdd(count_out_ptr, n_own);  // the queue size, for the statistics
        }
        if (kFuse) {
            // Fused form: the workgroup finishes the paths of its OWN segment -- no second launch, no dependency on any other
            // workgroup; workgroups that are done with their primary tiles run these latency-bound tails while others still trace
            // primaries.  Waves take 64 entries at a time from a counter in LDS.
            for (;;) {
                uint32_t w = 0;
                if (lane == 0) w = atomicAdd(&s_loop_next, 64u);
                w = __builtin_amdgcn_readfirstlane(w);
                if (w >= n_own) break;
                const uint32_t i = w + lane;
                if (i < n_own) {
                    PathState ps = load_path(qout, blockIdx.x * fc.seg_cap + i);
                    for (;;) {
                        float t;
                        uint32_t id;
                        closest_hit_any<kLds, StackT, kTex>(sv, nodes, sph, ids, ps.o, ps.d, 0.0f, kInf, stack, blockDim.x, t, id);
                        if (!shade_step<kMulti, kTex>(sv, pm, fp, scratch, out, ps, t, id)) break;
                        my_loop_rays++;
                    }
                }
            }
        }
    }
    if (kLoop || kIters > 1u || kFuse || kDI) {
        // rays traced in registers: wave reduce, then ONE atomic pair per workgroup (same-address device-scope atomics from
        // 8 XCDs serialise: one per wave made the non-persistent form of this kernel three times slower)
        unsigned long long total = my_rays, total2 = kFuse ? my_loop_rays : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            total += __shfl_down(total, off, 64);
            if (kFuse) total2 += __shfl_down(total2, off, 64);
        }
        __shared__ uint32_t s_wave_count2[kFusedThreads / 64];
        __syncthreads();  // s_wave_count is free again
        if (lane == 0) { s_wave_count[wave] = (uint32_t)total; if (kFuse) s_wave_count2[wave] = (uint32_t)total2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long sum = 0, sum2 = 0;
            for (uint32_t w = 0; w < (blockDim.x >> 6); w++) { sum += s_wave_count[w]; if (kFuse) sum2 += s_wave_count2[w]; }
            if (sum + sum2) atomicAdd(fc.tail_rays, sum + sum2);
            if (sum && !kLoop && !kDI) atomicAdd(fc.totals + 4, sum);  // running count of the rays a primary pass traced in registers (statistics)
        }
    }
}

// Launches the bounce_kernel instance for (kLds, StackT) that the run-time switches select.  One translation unit per
// (kLds, StackT) pair instantiates this (pt_bounce_*.hip); launch_bounce (pt_kernels.hip) picks the pair.
template <bool kLds, typename StackT>
hipError_t launch_bounce_for(const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const RayQueue& qout,
                             const Scratch& scratch, float4* out, const uint32_t* count_in, uint32_t* count_out, const FrameCounters& fc,
                             bool primary, bool loop, bool inline2, uint32_t threads, uint32_t grid, hipStream_t stream)
{
    const uint32_t lds = (sv.lds_scene ? scene_lds_bytes(sv.n_nodes, sv.n) : 0u) + threads * sv.stack_depth * (uint32_t)sizeof(StackT);
#define PT_BOUNCE8(P, LP, M, X, I, F, D)                                                                                   \
    do {                                                                                                                    \
        if (lds + kStaticLdsMargin > 65536u) (void)hipFuncSetAttribute((const void*)bounce_kernel<kLds, StackT, P, LP, M, X, I, F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((bounce_kernel<kLds, StackT, P, LP, M, X, I, F, D>), dim3(grid), dim3(threads), lds, stream, sv, pm, fp, qin, qout, scratch, out, count_in, count_out, fc); \
    } while (0)
/* the direct-illumination estimate (row N4) is made by the compacting primary pass */
#define PT_BOUNCE7(P, LP, M, X, I, F)                                                                                      \
    do { if (fp.di_enabled && P && !LP) PT_BOUNCE8(P, LP, M, X, I, F, (P && !LP)); else PT_BOUNCE8(P, LP, M, X, I, F, false); } while (0)
/* the fused form (the primary pass finishes its own segments) exists only for the compacting primary pass */
#define PT_BOUNCE6(P, LP, M, X, I)                                                                                         \
    do { if (fc.fuse_loop && P && !LP) PT_BOUNCE7(P, LP, M, X, I, (P && !LP)); else PT_BOUNCE7(P, LP, M, X, I, false); } while (0)
/* the in-register second bounce exists only for the 1-spp compacting primary pass */
#define PT_BOUNCE5(P, LP, M, X)                                                                                            \
    do { if (inline2 && P && !LP && !M) PT_BOUNCE6(P, LP, M, X, (P && !LP && !M)); else PT_BOUNCE6(P, LP, M, X, false); } while (0)
#define PT_BOUNCE4(P, LP, M)                                                                                               \
    do { if (sv.tex_maps) PT_BOUNCE5(P, LP, M, true); else PT_BOUNCE5(P, LP, M, false); } while (0)
#define PT_BOUNCE3(P, LP)                                                                                                  \
    do { if (fp.spp > 1) PT_BOUNCE4(P, LP, true); else PT_BOUNCE4(P, LP, false); } while (0)
    if (primary) { if (loop) PT_BOUNCE3(true, true); else PT_BOUNCE3(true, false); }
    else { if (loop) PT_BOUNCE3(false, true); else PT_BOUNCE3(false, false); }
#undef PT_BOUNCE3
#undef PT_BOUNCE4
#undef PT_BOUNCE5
#undef PT_BOUNCE6
#undef PT_BOUNCE7
#undef PT_BOUNCE8
    return hipGetLastError();
}

// the three instantiations (pt_bounce_lds16.hip, pt_bounce_g16.hip, pt_bounce_g32.hip).  An LDS-resident tree has fewer than
// 32767 nodes (64 KB hold ~700 spheres), so <true, uint32_t> does not exist.
#define PT_BOUNCE_LAUNCHER_ARGS const SceneView& sv, const PixelMap& pm, const FrameParams& fp, const RayQueue& qin, const RayQueue& qout, const Scratch& scratch, \
    float4* out, const uint32_t* count_in, uint32_t* count_out, const FrameCounters& fc, bool primary, bool loop, bool inline2, uint32_t threads, uint32_t grid, hipStream_t stream
hipError_t launch_bounce_lds16(PT_BOUNCE_LAUNCHER_ARGS);
hipError_t launch_bounce_g16(PT_BOUNCE_LAUNCHER_ARGS);
hipError_t launch_bounce_g32(PT_BOUNCE_LAUNCHER_ARGS);

}  // namespace pt
