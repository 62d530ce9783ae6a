// pt_lbvh_gpu.hip -- device LBVH builder: bounds -> 30-bit Morton keys -> radix sort (hipCUB / rocPRIM) -> Karras 2012
// radix tree -> depth -> bottom-up AABBs in `depth` communication-free passes.  Produces exactly the tree of the host builder (pt_lbvh.cpp): every floating
// point step is the same individually rounded fp32 operation, min/max are exact, and the keys are unique.
// Replaces Scene::CreateAccelerationStructures (Source/Scene.ixx:225-284), which the reference re-runs every frame while
// the physics is live (Source/App.cpp:605-608).
#include "pt_lbvh_gpu.h"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <new>

namespace pt {

namespace {

// order-preserving float <-> uint mapping for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

// header[0..2] centroid min, [3..5] centroid max, [6..8] bounds min, [9..11] bounds max (ordered uints), [12] depth
constexpr int kHdrWords = 16;

__global__ void init_header_kernel(uint32_t* hdr)
{
    const uint32_t i = threadIdx.x;
    if (i < 12) hdr[i] = ((i / 3) & 1) ? 0u : 0xFFFFFFFFu;  // mins start at +max, maxes at 0
    if (i == 12) hdr[12] = 0u;
}

// Scene bounds.  Per-wave shuffle reduction, then per-block through LDS: 12 atomics per BLOCK on the 12 header words (one set
// per wave -- 196 K same-address atomics at 2^20 spheres -- cost 2.2 ms; this costs ~10 us).
__global__ __launch_bounds__(256) void bounds_kernel(const float4* __restrict__ sph, uint32_t n, uint32_t* __restrict__ hdr)
{
    __shared__ float s_red[4][12];
    float v[12];  // cmin[3], cmax[3], bmin[3], bmax[3]
#pragma unroll
    for (int a = 0; a < 3; a++) { v[a] = INFINITY; v[3 + a] = -INFINITY; v[6 + a] = INFINITY; v[9 + a] = -INFINITY; }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 s = sph[i];
        const float c[3] = { s.x, s.y, s.z };
#pragma unroll
        for (int a = 0; a < 3; a++) {
            v[a] = fminf(v[a], c[a]); v[3 + a] = fmaxf(v[3 + a], c[a]);
            v[6 + a] = fminf(v[6 + a], c[a] - s.w); v[9 + a] = fmaxf(v[9 + a], c[a] + s.w);
        }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const bool is_min = (k / 3) % 2 == 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_down(v[k], off, 64);
            v[k] = is_min ? fminf(v[k], o) : fmaxf(v[k], o);
        }
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) s_red[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int k = threadIdx.x;
        const bool is_min = (k / 3) % 2 == 0;
        float r = s_red[0][k];
        for (uint32_t w = 1; w < (blockDim.x >> 6); w++) r = is_min ? fminf(r, s_red[w][k]) : fmaxf(r, s_red[w][k]);
        if (is_min) atomicMin(&hdr[k], f2ord(r)); else atomicMax(&hdr[k], f2ord(r));
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t quant10(float v)
{
    float s = v * 1024.0f;
    s = s < 0.0f ? 0.0f : (s > 1023.0f ? 1023.0f : s);
    return (uint32_t)s;
}

__global__ void morton_kernel(const float4* __restrict__ sph, uint32_t n, const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ keys)
{
    float cmin[3], inv[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        cmin[a] = ord2f(hdr[a]);
        const float e = ord2f(hdr[3 + a]) - cmin[a];
        inv[a] = e > 0.0f ? 1.0f / e : 0.0f;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 s = sph[i];
        const uint32_t code = (expand10(quant10((s.x - cmin[0]) * inv[0])) << 2) | (expand10(quant10((s.y - cmin[1]) * inv[1])) << 1)
                              | expand10(quant10((s.z - cmin[2]) * inv[2]));
        keys[i] = ((unsigned long long)code << 32) | i;
    }
}

__global__ void gather_kernel(const float4* __restrict__ sph, const unsigned long long* __restrict__ keys, uint32_t n, float4* __restrict__ sorted,
                              uint32_t* __restrict__ sorted_id)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t)(keys[k] & 0xFFFFFFFFull);
        sorted_id[k] = id;
        sorted[k] = sph[id];
    }
}

__global__ void gather_by_id_kernel(const float4* __restrict__ sph, const uint32_t* __restrict__ sorted_id, uint32_t n, float4* __restrict__ sorted)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) sorted[k] = sph[sorted_id[k]];
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras 2012, one thread per internal node; also records each child's parent
__global__ void hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, PtBvhNode* __restrict__ nodes, int* __restrict__ leaf_parent)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) > 0 ? 1 : -1;
        const int dmin = delta(keys, n, i, i - d);
        int lmax = 2;
        while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = delta(keys, n, i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        const int gamma = i + s * d + min(d, 0);
        const int lo = min(i, j), hi = max(i, j);
        const int c0 = (lo == gamma) ? ~gamma : gamma;
        const int c1 = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
        nodes[i].child0 = c0;
        nodes[i].child1 = c1;
        nodes[i]._pad = 0;
        if (i == 0) nodes[0].parent = -1;
        if (c0 >= 0) nodes[c0].parent = i; else leaf_parent[~c0] = i;
        if (c1 >= 0) nodes[c1].parent = i; else leaf_parent[~c1] = i;
    }
}

// Bottom-up AABBs without inter-thread communication inside a launch.  A node's record holds its CHILDREN's boxes
// (PtBvhNode layout); it can be written once both children are complete (a leaf is complete from the start; an internal
// child is complete once ITS record is written, and its own box is the union of the two boxes stored there).
//   level[i] = the pass in which node i was completed (0xFFFFFFFF = not yet); a child counts as complete only if its
//   level is < the current pass, i.e. it was written by an EARLIER launch (n > 1024) or before the last barrier (n <= 1024),
//   so a racing same-pass write is never consumed.  Passes needed = tree depth.
// This replaces one-thread-per-leaf climbing with atomic arrival counters: the agent-scope release / acquire fences that
// scheme needs write back and invalidate the per-XCD L2 on this GPU -- 7.2 ms at 2^20 spheres, 35 us at 441.
__device__ __forceinline__ void leaf_box(const float4 s, float pad, float lo[3], float hi[3])
{
    lo[0] = s.x - s.w - pad; lo[1] = s.y - s.w - pad; lo[2] = s.z - s.w - pad;
    hi[0] = s.x + s.w + pad; hi[1] = s.y + s.w + pad; hi[2] = s.z + s.w + pad;
}

__device__ __forceinline__ void child_box(const float4* __restrict__ sorted, const PtBvhNode* nodes, int c, float pad, float lo[3], float hi[3])
{
    if (c < 0) {
        leaf_box(sorted[~c], pad, lo, hi);
    } else {
        const PtBvhNode* nd = &nodes[c];
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = fminf(nd->lo0[a], nd->lo1[a]); hi[a] = fmaxf(nd->hi0[a], nd->hi1[a]); }
    }
}

__device__ __forceinline__ float refit_padding(const uint32_t* __restrict__ hdr_ro)
{
    // padding = 2^-17 * max |coordinate| of the scene bounds (lbvh_padding in pt_lbvh.cpp)
    float smax = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) smax = fmaxf(smax, fmaxf(fabsf(ord2f(hdr_ro[6 + a])), fabsf(ord2f(hdr_ro[9 + a]))));
    return smax * 7.62939453125e-06f;
}

__device__ __forceinline__ void write_node_boxes(const float4* __restrict__ sorted, PtBvhNode* nodes, int i, int c0, int c1, float pad)
{
    float lo[3], hi[3];
    PtBvhNode* nd = &nodes[i];
    child_box(sorted, nodes, c0, pad, lo, hi);
#pragma unroll
    for (int a = 0; a < 3; a++) { nd->lo0[a] = lo[a]; nd->hi0[a] = hi[a]; }
    child_box(sorted, nodes, c1, pad, lo, hi);
#pragma unroll
    for (int a = 0; a < 3; a++) { nd->lo1[a] = lo[a]; nd->hi1[a] = hi[a]; }
}

constexpr int kRefitSmall = 1024;  // up to this many spheres one workgroup does every pass (barriers instead of launches)

__global__ __launch_bounds__(kRefitSmall) void refit_small_kernel(const float4* __restrict__ sorted, int n, PtBvhNode* nodes,
                                                                  uint32_t* __restrict__ hdr)
{
    __shared__ uint32_t s_level[kRefitSmall];
    const float pad = refit_padding(hdr);
    const int i = threadIdx.x;
    const bool is_node = i < n - 1;
    int c0 = 0, c1 = 0;
    if (is_node) { c0 = nodes[i].child0; c1 = nodes[i].child1; }
    s_level[i] = 0xFFFFFFFFu;
    __syncthreads();
    bool done = !is_node;
    for (uint32_t pass = 1; pass <= 64u; pass++) {
        if (!done) {
            const bool r0 = c0 < 0 || s_level[c0] < pass, r1 = c1 < 0 || s_level[c1] < pass;
            if (r0 && r1) {
                write_node_boxes(sorted, nodes, i, c0, c1, pad);
                s_level[i] = pass;
                done = true;
            }
        }
        __syncthreads();  // the records written in this pass are visible to the whole workgroup
        if (s_level[0] != 0xFFFFFFFFu) break;  // the root is complete (it is written once, so every thread agrees)
    }
    if (i == 0) hdr[12] = s_level[0];  // tree depth = the pass that completed the root
}

// The whole per-frame update of a small scene (n <= kRefitSmall) in ONE launch: the new spheres are read from `src` -- which may be
// the caller's pinned staging buffer in host memory -- and copied to the device array the shading reads (`sph_copy`), the scene bounds
// are reduced in LDS, the Morton-ordered copy is gathered, and the boxes are refitted bottom-up as in refit_small_kernel (the leaves
// come from LDS: `sorted` is written by this very workgroup).  Replaces upload + init_header + bounds + gather + refit_small: five
// dependent operations at the head of every frame of an animated scene.  Same arithmetic, same results.
__global__ __launch_bounds__(kRefitSmall) void refit_fused_small_kernel(const float4* src, float4* __restrict__ sph_copy, const uint32_t* __restrict__ sorted_id,
                                                                        float4* __restrict__ sorted, int n, PtBvhNode* nodes, uint32_t* hdr)
{
    __shared__ uint32_t s_level[kRefitSmall];
    __shared__ float4 s_sph[kRefitSmall];
    __shared__ float4 s_sorted[kRefitSmall];
    __shared__ uint32_t s_hdr[16];
    const int i = threadIdx.x;
    if (i < 12) s_hdr[i] = ((i / 3) & 1) ? 0u : 0xFFFFFFFFu;  // init_header_kernel
    float v[12];
#pragma unroll
    for (int a = 0; a < 3; a++) { v[a] = INFINITY; v[3 + a] = -INFINITY; v[6 + a] = INFINITY; v[9 + a] = -INFINITY; }
    if (i < n) {
        const float4 sp = src[i];
        s_sph[i] = sp;
        sph_copy[i] = sp;
        const float c[3] = { sp.x, sp.y, sp.z };
#pragma unroll
        for (int a = 0; a < 3; a++) { v[a] = c[a]; v[3 + a] = c[a]; v[6 + a] = c[a] - sp.w; v[9 + a] = c[a] + sp.w; }  // bounds_kernel
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const bool is_min = (k / 3) % 2 == 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_down(v[k], off, 64);
            v[k] = is_min ? fminf(v[k], o) : fmaxf(v[k], o);
        }
    }
    __syncthreads();  // s_hdr initialised, s_sph complete
    if ((i & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            if ((k / 3) % 2 == 0) atomicMin(&s_hdr[k], f2ord(v[k])); else atomicMax(&s_hdr[k], f2ord(v[k]));
        }
    }
    if (i < n) { const float4 g = s_sph[sorted_id[i]]; s_sorted[i] = g; sorted[i] = g; }  // gather_by_id_kernel
    const bool is_node = i < n - 1;
    int c0 = 0, c1 = 0;
    if (is_node) { c0 = nodes[i].child0; c1 = nodes[i].child1; }
    s_level[i] = 0xFFFFFFFFu;
    __syncthreads();  // bounds reduced, s_sorted complete
    if (i < 12) hdr[i] = s_hdr[i];
    const float pad = refit_padding(s_hdr);
    bool done = !is_node;
    for (uint32_t pass = 1; pass <= 64u; pass++) {
        if (!done) {
            const bool r0 = c0 < 0 || s_level[c0] < pass, r1 = c1 < 0 || s_level[c1] < pass;
            if (r0 && r1) {
                write_node_boxes(s_sorted, nodes, i, c0, c1, pad);
                s_level[i] = pass;
                done = true;
            }
        }
        __syncthreads();
        if (s_level[0] != 0xFFFFFFFFu) break;
    }
    if (i == 0) hdr[12] = s_level[0];
}

__global__ void refit_pass_kernel(const float4* __restrict__ sorted, int n, PtBvhNode* nodes, uint32_t* level, uint32_t pass,
                                  const uint32_t* __restrict__ hdr_ro)
{
    const float pad = refit_padding(hdr_ro);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        if (level[i] != 0xFFFFFFFFu) continue;
        const int c0 = nodes[i].child0, c1 = nodes[i].child1;
        const bool r0 = c0 < 0 || level[c0] < pass, r1 = c1 < 0 || level[c1] < pass;
        if (!(r0 && r1)) continue;
        write_node_boxes(sorted, nodes, i, c0, c1, pad);
        level[i] = pass;
    }
}

// boxes of every node record; `level` = n uint32 of scratch; depth = tree depth (passes) for n > kRefitSmall
static hipError_t launch_refit(const float4* sorted, uint32_t n, PtBvhNode* nodes, uint32_t* level, uint32_t* hdr, uint32_t depth, hipStream_t stream)
{
    if (n <= 1) return hipSuccess;
    if (n <= (uint32_t)kRefitSmall) {
        hipLaunchKernelGGL(refit_small_kernel, dim3(1), dim3(kRefitSmall), 0, stream, sorted, (int)n, nodes, hdr);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(level, 0xFF, (size_t)n * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t threads = 256, grid = (n + threads - 1) / threads < 8192u ? (n + threads - 1) / threads : 8192u;
    for (uint32_t pass = 1; pass <= depth; pass++)
        hipLaunchKernelGGL(refit_pass_kernel, dim3(grid), dim3(threads), 0, stream, sorted, (int)n, nodes, level, pass, hdr);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void depth_kernel(const PtBvhNode* __restrict__ nodes, const int* __restrict__ leaf_parent, int n, uint32_t* __restrict__ hdr)
{
    __shared__ uint32_t s_best[4];
    uint32_t best = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t d = 0;
        for (int p = leaf_parent[k]; p >= 0; p = nodes[p].parent) d++;
        best = max(best, d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, (uint32_t)__shfl_down(best, off, 64));
    if ((threadIdx.x & 63u) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per workgroup (same-address atomics from every wave serialise)
        for (uint32_t w = 1; w < (blockDim.x >> 6); w++) best = max(best, s_best[w]);
        if (best) atomicMax(&hdr[12], best);
    }
}

}  // namespace

struct LbvhGpu {
    unsigned long long* keys_in = nullptr;
    unsigned long long* keys_out = nullptr;
    int* leaf_parent = nullptr;
    uint32_t* flags = nullptr;
    uint32_t* hdr = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    size_t cap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

bool lbvh_gpu_available() { return true; }

LbvhGpu* lbvh_gpu_create()
{
    LbvhGpu* b = new (std::nothrow) LbvhGpu();
    if (!b) return nullptr;
    if (hipMalloc(&b->hdr, kHdrWords * sizeof(uint32_t)) != hipSuccess || hipEventCreate(&b->e0) != hipSuccess || hipEventCreate(&b->e1) != hipSuccess) {
        lbvh_gpu_destroy(b);
        return nullptr;
    }
    return b;
}

static void free_buffers(LbvhGpu* b)
{
    if (b->keys_in) (void)hipFree(b->keys_in);
    if (b->keys_out) (void)hipFree(b->keys_out);
    if (b->leaf_parent) (void)hipFree(b->leaf_parent);
    if (b->flags) (void)hipFree(b->flags);
    if (b->sort_tmp) (void)hipFree(b->sort_tmp);
    b->keys_in = b->keys_out = nullptr; b->leaf_parent = nullptr; b->flags = nullptr; b->sort_tmp = nullptr;
    b->cap = 0; b->sort_tmp_bytes = 0;
}

void lbvh_gpu_destroy(LbvhGpu* b)
{
    if (!b) return;
    free_buffers(b);
    if (b->hdr) (void)hipFree(b->hdr);
    if (b->e0) (void)hipEventDestroy(b->e0);
    if (b->e1) (void)hipEventDestroy(b->e1);
    delete b;
}

#define LB_CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

static hipError_t ensure_capacity(LbvhGpu* b, uint32_t n, hipStream_t stream)
{
    if (n <= b->cap) return hipSuccess;
    LB_CK(hipStreamSynchronize(stream));
    free_buffers(b);
    LB_CK(hipMalloc(&b->keys_in, (size_t)n * sizeof(unsigned long long)));
    LB_CK(hipMalloc(&b->keys_out, (size_t)n * sizeof(unsigned long long)));
    LB_CK(hipMalloc(&b->leaf_parent, (size_t)n * sizeof(int)));
    LB_CK(hipMalloc(&b->flags, (size_t)n * sizeof(uint32_t)));
    size_t tmp = 0;
    LB_CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, b->keys_in, b->keys_out, (int)n, 0, 62, stream));
    LB_CK(hipMalloc(&b->sort_tmp, tmp ? tmp : 16));
    b->sort_tmp_bytes = tmp;
    b->cap = n;
    return hipSuccess;
}

static void read_info(const uint32_t* hdr, uint32_t depth, float ms, LbvhGpuInfo* info)
{
    info->build_ms = ms;
    info->depth = depth;
    float smax = 0.0f;
    for (int a = 0; a < 3; a++) {
        info->bounds_min[a] = ord2f(hdr[6 + a]);
        info->bounds_max[a] = ord2f(hdr[9 + a]);
        smax = std::fmax(smax, std::fmax(std::fabs(info->bounds_min[a]), std::fabs(info->bounds_max[a])));
    }
    info->pad = smax * 7.62939453125e-06f;
}

hipError_t lbvh_gpu_build(LbvhGpu* b, const float4* d_sph, uint32_t n, PtBvhNode* d_nodes, float4* d_sorted, uint32_t* d_sorted_id,
                          hipStream_t stream, LbvhGpuInfo* info)
{
    if (!b || !d_sph || n == 0 || !d_sorted || !d_sorted_id || !info) return hipErrorInvalidValue;
    LB_CK(ensure_capacity(b, n, stream));
    const uint32_t threads = 256;
    const uint32_t grid = (n + threads - 1) / threads < 4096u ? (n + threads - 1) / threads : 4096u;
    const uint32_t red_grid = grid < 512u ? grid : 512u;
    LB_CK(hipEventRecord(b->e0, stream));
    hipLaunchKernelGGL(init_header_kernel, dim3(1), dim3(64), 0, stream, b->hdr);
    hipLaunchKernelGGL(bounds_kernel, dim3(red_grid), dim3(threads), 0, stream, d_sph, n, b->hdr);
    hipLaunchKernelGGL(morton_kernel, dim3(grid), dim3(threads), 0, stream, d_sph, n, b->hdr, b->keys_in);
    size_t tmp = b->sort_tmp_bytes;
    LB_CK(hipcub::DeviceRadixSort::SortKeys(b->sort_tmp, tmp, b->keys_in, b->keys_out, (int)n, 0, 62, stream));
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(threads), 0, stream, d_sph, b->keys_out, n, d_sorted, d_sorted_id);
    uint32_t hdr[kHdrWords];
    if (n > 1) {
        hipLaunchKernelGGL(hierarchy_kernel, dim3(grid), dim3(threads), 0, stream, b->keys_out, (int)n, d_nodes, b->leaf_parent);
        uint32_t depth = 0;
        if (n > (uint32_t)kRefitSmall) {
            // the number of refit passes is the tree depth: measure it first (leaf -> root walks), one small read-back
            hipLaunchKernelGGL(depth_kernel, dim3(grid), dim3(threads), 0, stream, d_nodes, b->leaf_parent, (int)n, b->hdr);
            LB_CK(hipMemcpyAsync(hdr, b->hdr, sizeof hdr, hipMemcpyDeviceToHost, stream));
            LB_CK(hipStreamSynchronize(stream));
            depth = hdr[12];
        }
        LB_CK(launch_refit(d_sorted, n, d_nodes, b->flags, b->hdr, depth, stream));
    }
    LB_CK(hipGetLastError());
    LB_CK(hipEventRecord(b->e1, stream));
    LB_CK(hipMemcpyAsync(hdr, b->hdr, sizeof hdr, hipMemcpyDeviceToHost, stream));
    LB_CK(hipStreamSynchronize(stream));
    float ms = 0;
    LB_CK(hipEventElapsedTime(&ms, b->e0, b->e1));
    read_info(hdr, n > 1 ? hdr[12] : 0u, ms, info);
    return hipSuccess;
}

// A topology built elsewhere (the host SAH builder): upload the records' links and the leaf order, then let the refit passes
// -- topology-agnostic -- compute every box, exactly as a later lbvh_gpu_refit will.  `depth` bounds the passes for
// n > kRefitSmall (the single-workgroup form finds the depth itself).
hipError_t lbvh_gpu_adopt(LbvhGpu* b, const float4* d_sph, uint32_t n, const PtBvhNode* h_nodes, const uint32_t* h_sorted_id, uint32_t depth,
                          PtBvhNode* d_nodes, float4* d_sorted, uint32_t* d_sorted_id, hipStream_t stream, LbvhGpuInfo* info)
{
    if (!b || !d_sph || n == 0 || !d_sorted || !d_sorted_id || !h_sorted_id || !info || (n > 1 && (!h_nodes || !d_nodes))) return hipErrorInvalidValue;
    LB_CK(ensure_capacity(b, n, stream));
    const uint32_t threads = 256;
    const uint32_t grid = (n + threads - 1) / threads < 4096u ? (n + threads - 1) / threads : 4096u;
    const uint32_t red_grid = grid < 512u ? grid : 512u;
    LB_CK(hipEventRecord(b->e0, stream));
    if (n > 1) LB_CK(hipMemcpyAsync(d_nodes, h_nodes, (size_t)(n - 1) * sizeof(PtBvhNode), hipMemcpyHostToDevice, stream));
    LB_CK(hipMemcpyAsync(d_sorted_id, h_sorted_id, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(init_header_kernel, dim3(1), dim3(64), 0, stream, b->hdr);
    hipLaunchKernelGGL(bounds_kernel, dim3(red_grid), dim3(threads), 0, stream, d_sph, n, b->hdr);
    hipLaunchKernelGGL(gather_by_id_kernel, dim3(grid), dim3(threads), 0, stream, d_sph, d_sorted_id, n, d_sorted);
    LB_CK(launch_refit(d_sorted, n, d_nodes, b->flags, b->hdr, depth, stream));
    LB_CK(hipGetLastError());
    LB_CK(hipEventRecord(b->e1, stream));
    uint32_t hdr[kHdrWords];
    LB_CK(hipMemcpyAsync(hdr, b->hdr, sizeof hdr, hipMemcpyDeviceToHost, stream));
    LB_CK(hipStreamSynchronize(stream));
    float ms = 0;
    LB_CK(hipEventElapsedTime(&ms, b->e0, b->e1));
    read_info(hdr, n > 1 ? depth : 0u, ms, info);
    return hipSuccess;
}

hipError_t lbvh_gpu_refit(LbvhGpu* b, const float4* d_sph, uint32_t n, PtBvhNode* d_nodes, float4* d_sorted, const uint32_t* d_sorted_id,
                          uint32_t* d_flags, uint32_t* d_hdr, uint32_t depth, hipStream_t stream)
{
    if (!b || !d_sph || n == 0 || !d_sorted || !d_sorted_id || !d_flags || !d_hdr || n > b->cap) return hipErrorInvalidValue;
    const uint32_t threads = 256;
    const uint32_t grid = (n + threads - 1) / threads < 4096u ? (n + threads - 1) / threads : 4096u;
    const uint32_t red_grid = grid < 512u ? grid : 512u;
    hipLaunchKernelGGL(init_header_kernel, dim3(1), dim3(64), 0, stream, d_hdr);
    hipLaunchKernelGGL(bounds_kernel, dim3(red_grid), dim3(threads), 0, stream, d_sph, n, d_hdr);  // the padding follows the new bounds
    hipLaunchKernelGGL(gather_by_id_kernel, dim3(grid), dim3(threads), 0, stream, d_sph, d_sorted_id, n, d_sorted);
    LB_CK(launch_refit(d_sorted, n, d_nodes, d_flags, d_hdr, depth, stream));
    return hipGetLastError();
}

bool lbvh_gpu_refit_fused_possible(uint32_t n) { return n > 1 && n <= (uint32_t)kRefitSmall; }

// lbvh_gpu_refit for small scenes with the upload folded in: src = the new spheres (device memory or pinned host memory as the device
// sees it), d_sph = the device array they are copied to
hipError_t lbvh_gpu_refit_fused(LbvhGpu* b, const float4* src, float4* d_sph, uint32_t n, PtBvhNode* d_nodes, float4* d_sorted, const uint32_t* d_sorted_id,
                                uint32_t* d_hdr, hipStream_t stream)
{
    if (!b || !src || !d_sph || !lbvh_gpu_refit_fused_possible(n) || !d_sorted || !d_sorted_id || !d_hdr || n > b->cap) return hipErrorInvalidValue;
    hipLaunchKernelGGL(refit_fused_small_kernel, dim3(1), dim3(kRefitSmall), 0, stream, src, d_sph, d_sorted_id, d_sorted, (int)n, d_nodes, d_hdr);
    return hipGetLastError();
}

// ---- 4-wide, quantised view of the binary tree (global-memory scenes; DESIGN.md "Wide nodes") ------------------------------------
// A binary node at EVEN depth becomes a wide node whose children are its grandchildren (a leaf child stays as it is): the
// odd-depth nodes are absorbed, every second level of dependent fetches disappears -- and the four child boxes are stored as 8-bit
// offsets on a per-node power-of-two grid, so a visit reads ONE 64-byte sector (the binary walk reads one per level: half the
// sectors per ray, which is what bounds the 2^20-sphere scene).  Wide node i lives at index i of its own array (only even-depth
// slots are used), 16 dwords:
//   [0..2] grid origin (the union box's lower corner)   [3] three biased exponents (x | y << 8 | z << 16): cell = 2^(e - 127)
//   [4..6] lower x / y / z plane of children 0..3, one byte each   [7..9] upper planes   [10..13] child references (>= 0: wide
//   node index, < 0: leaf ~k, 0x80000000: empty)   [14..15] unused
// A plane decodes as fma(byte, cell, origin) -- the very expression the builder checks here, on the same hardware: lower planes
// are rounded down and upper planes up until the decoded box contains the binary record's box, so conservativeness carries over.
constexpr int kEmptyChild = (int)0x80000000;

__device__ __forceinline__ float wide_cell(float extent, uint32_t& biased_exp)
{
    // smallest power of two c with 254 * c >= extent (one step of headroom for the outward rounding below)
    int e = 0;
    const float m = frexpf(extent * (1.0f / 254.0f), &e);  // extent / 254 = m * 2^e, m in [0.5, 1)
    (void)m;
    if (!(extent > 0.0f)) e = -100;
    if (e < -120) e = -120;
    biased_exp = (uint32_t)(e + 127);
    return __uint_as_float(biased_exp << 23);
}

__global__ void collapse4_kernel(const PtBvhNode* __restrict__ nodes, uint32_t n_nodes, uint4* __restrict__ wide)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += gridDim.x * blockDim.x) {
        uint32_t depth = 0;
        for (int p = nodes[i].parent; p >= 0; p = nodes[p].parent) depth++;
        if (depth & 1u) continue;
        float lo[4][3], hi[4][3];
        int ref[4];
        int k = 0;
        auto put = [&](const float* l, const float* h, int c) {
            for (int a = 0; a < 3; a++) { lo[k][a] = l[a]; hi[k][a] = h[a]; }
            ref[k++] = c;
        };
        const PtBvhNode nd = nodes[i];
        const int cs[2] = { nd.child0, nd.child1 };
        for (int side = 0; side < 2; side++) {
            const int c = cs[side];
            if (c < 0) {
                put(side ? nd.lo1 : nd.lo0, side ? nd.hi1 : nd.hi0, c);
            } else {  // odd-depth internal child: absorbed
                const PtBvhNode ch = nodes[c];
                put(ch.lo0, ch.hi0, ch.child0);
                put(ch.lo1, ch.hi1, ch.child1);
            }
        }
        const int n_children = k;
        float origin[3], cell[3];
        uint32_t exps = 0, planes[6] = { 0, 0, 0, 0, 0, 0 };
        for (int a = 0; a < 3; a++) {
            float mn = lo[0][a], mx = hi[0][a];
            for (int c = 1; c < n_children; c++) { mn = fminf(mn, lo[c][a]); mx = fmaxf(mx, hi[c][a]); }
            origin[a] = mn;
            uint32_t be;
            cell[a] = wide_cell(mx - mn, be);
            exps |= be << (8 * a);
            const float inv = 1.0f / cell[a];  // exact: a power of two
            for (int c = 0; c < n_children; c++) {
                int ql = (int)floorf((lo[c][a] - mn) * inv), qh = (int)ceilf((hi[c][a] - mn) * inv);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (ql > 0 && __fmaf_rn((float)ql, cell[a], mn) > lo[c][a]) ql--;
                while (qh < 255 && __fmaf_rn((float)qh, cell[a], mn) < hi[c][a]) qh++;
                planes[a] |= (uint32_t)ql << (8 * c);
                planes[3 + a] |= (uint32_t)qh << (8 * c);
            }
        }
        for (int c = n_children; c < 4; c++) ref[c] = kEmptyChild;
        uint4* w = wide + (size_t)i * 4u;
        w[0] = make_uint4(__float_as_uint(origin[0]), __float_as_uint(origin[1]), __float_as_uint(origin[2]), exps);
        w[1] = make_uint4(planes[0], planes[1], planes[2], planes[3]);
        w[2] = make_uint4(planes[4], planes[5], (uint32_t)ref[0], (uint32_t)ref[1]);
        w[3] = make_uint4((uint32_t)ref[2], (uint32_t)ref[3], 0u, 0u);
    }
}

hipError_t lbvh_gpu_collapse4(const PtBvhNode* d_nodes, uint32_t n_nodes, float4* d_wide, hipStream_t stream)
{
    if (n_nodes == 0) return hipSuccess;
    if (!d_nodes || !d_wide) return hipErrorInvalidValue;
    const uint32_t grid = (n_nodes + 255u) / 256u < 4096u ? (n_nodes + 255u) / 256u : 4096u;
    hipLaunchKernelGGL(collapse4_kernel, dim3(grid), dim3(256), 0, stream, d_nodes, n_nodes, reinterpret_cast<uint4*>(d_wide));
    return hipGetLastError();
}

}  // namespace pt
