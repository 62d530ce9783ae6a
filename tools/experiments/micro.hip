// micro-benchmarks for the traverse launch floor (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while(0)

__global__ void k_empty(const uint32_t* count_ptr) { if (blockIdx.x * blockDim.x >= *count_ptr) return; }

__global__ void k_stage(const float4* src, uint32_t n_vec, const uint32_t* count_ptr, float4* sink)
{
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * blockDim.x >= count) return;
    for (uint32_t i = threadIdx.x; i < n_vec; i += blockDim.x) smem[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = smem[(blockIdx.x * 7) % n_vec];
}

// unrolled staging: all loads issued before any store
template <int N>
__global__ void k_stage_unrolled(const float4* src, uint32_t n_vec, const uint32_t* count_ptr, float4* sink)
{
    extern __shared__ float4 smem[];
    const uint32_t count = *count_ptr;
    if (blockIdx.x * blockDim.x >= count) return;
    float4 v[N];
#pragma unroll
    for (int k = 0; k < N; k++) { uint32_t i = threadIdx.x + k * blockDim.x; v[k] = i < n_vec ? src[i] : make_float4(0,0,0,0); }
#pragma unroll
    for (int k = 0; k < N; k++) { uint32_t i = threadIdx.x + k * blockDim.x; if (i < n_vec) smem[i] = v[k]; }
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = smem[(blockIdx.x * 7) % n_vec];
}

int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    uint32_t* d_count; CK(hipMalloc(&d_count, 4));
    const uint32_t n_vec = 2368;
    float4* d_src; CK(hipMalloc(&d_src, n_vec * 16)); CK(hipMemset(d_src, 0, n_vec * 16));
    float4* d_sink; CK(hipMalloc(&d_sink, 4096 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& f) {
        for (int i = 0; i < 5; i++) f();
        CK(hipStreamSynchronize(st));
        const int reps = 50;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-60s %8.2f us/launch\n", name, ms * 1e3 / reps);
    };
    for (uint32_t count : {0u, 1932u, 31452u, 1006940u}) {
        CK(hipMemcpy(d_count, &count, 4, hipMemcpyHostToDevice));
        printf("--- count = %u\n", count);
        for (uint32_t grid : {4u, 64u, 512u, 2048u}) {
            for (uint32_t lds : {0u, 57344u}) {
                char nm[128];
                snprintf(nm, sizeof nm, "empty grid=%u block=512 lds=%u", grid, lds);
                timeit(nm, [&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), lds, st, d_count); });
            }
            char nm[128];
            snprintf(nm, sizeof nm, "stage(loop) grid=%u block=512 lds=57344", grid);
            timeit(nm, [&] { hipLaunchKernelGGL(k_stage, dim3(grid), dim3(512), 57344, st, d_src, n_vec, d_count, d_sink); });
            snprintf(nm, sizeof nm, "stage(unrolled5) grid=%u block=512 lds=57344", grid);
            timeit(nm, [&] { hipLaunchKernelGGL(k_stage_unrolled<5>, dim3(grid), dim3(512), 57344, st, d_src, n_vec, d_count, d_sink); });
        }
    }
    return 0;
}
