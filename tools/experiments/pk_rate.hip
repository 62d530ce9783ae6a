// pk_rate.hip -- issue cost of the packed fp32 and three-operand min / max instructions next to v_fma_f32 on one gfx950 SIMD (4 waves per SIMD,
// 8 independent chains per lane): what a slab test written with v_pk_fma_f32 could gain.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_rate tools/experiments/pk_rate.hip && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define BODY(NAME, T, INIT, EXPR, FOLD)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int n_outer, float a, float b)                              \
    {                                                                                                                      \
        T x[8];                                                                                                            \
        _Pragma("unroll") for (int c = 0; c < 8; c++) x[c] = INIT;                                                         \
        for (int i = 0; i < n_outer; i++) {                                                                                \
            _Pragma("unroll") for (int k = 0; k < 30; k++) {                                                               \
                _Pragma("unroll") for (int c = 0; c < 8; c++) { T v = x[c]; x[c] = (EXPR); }                               \
            }                                                                                                              \
        }                                                                                                                  \
        float s = 0;                                                                                                       \
        _Pragma("unroll") for (int c = 0; c < 8; c++) s += FOLD;                                                           \
        if (s == 123.456f) out[0] = 1;                                                                                     \
    }

__device__ __forceinline__ float fma_asm(float v, float a, float b) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_fma_asm(f2 v, f2 a, f2 b) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_mul_asm(f2 v, f2 a) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ f2 pk_add_asm(f2 v, f2 a) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ float max_asm(float v, float a) { float r; asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ float max3_asm(float v, float a, float b) { float r; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fma_then_max(float v, float a, float b) { return max_asm(fma_asm(v, a, b), b); }

BODY(k_fma, float, (float)threadIdx.x + c, fma_asm(v, a, b), x[c])
BODY(k_pk_fma, f2, (f2{ (float)threadIdx.x + c, (float)c }), pk_fma_asm(v, (f2{ a, a }), (f2{ b, b })), x[c].x + x[c].y)
BODY(k_pk_mul, f2, (f2{ (float)threadIdx.x + c, (float)c }), pk_mul_asm(v, (f2{ a, a })), x[c].x + x[c].y)
BODY(k_pk_add, f2, (f2{ (float)threadIdx.x + c, (float)c }), pk_add_asm(v, (f2{ a, a })), x[c].x + x[c].y)
BODY(k_max, float, (float)threadIdx.x + c, max_asm(v, a), x[c])
BODY(k_max3, float, (float)threadIdx.x + c, max3_asm(v, a, b), x[c])
BODY(k_fma_max, float, (float)threadIdx.x + c, fma_then_max(v, a, b), x[c])

template <typename K>
static void run(const char* name, K kernel, unsigned* d, int num_cus, int per_iter)
{
    const int n_outer = 1 << 12;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kernel, dim3(num_cus * 4 * 4), dim3(256), 0, 0, d, n_outer, 1.0000001f, 1e-9f);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double insts_per_wave = (double)n_outer * 240 * per_iter;
    const double wave_insts_per_simd = insts_per_wave * 4 * 4;  // 4 waves per SIMD x 4 rounds
    std::printf("%-28s %8.3f ms -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, best, best * 1e6 / wave_insts_per_simd * 2.4);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int num_cus = prop.multiProcessorCount;
    unsigned* d; hipMalloc(&d, 4);
    run("v_fma_f32", k_fma, d, num_cus, 1);
    run("v_pk_fma_f32", k_pk_fma, d, num_cus, 1);
    run("v_pk_mul_f32", k_pk_mul, d, num_cus, 1);
    run("v_pk_add_f32", k_pk_add, d, num_cus, 1);
    run("v_max_f32", k_max, d, num_cus, 1);
    run("v_max3_f32", k_max3, d, num_cus, 1);
    run("v_fma_f32 + v_max_f32 (2)", k_fma_max, d, num_cus, 2);
    return 0;
}
