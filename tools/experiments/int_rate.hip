// int_rate.hip -- issue cost of the integer and conversion instructions the per-pixel RNG (Rng::Hash: two v_mul_lo_u32 per draw) and the
// IEEE div / sqrt expansions are made of, next to v_fma_f32, on one gfx950 SIMD (4 waves per SIMD, 8 independent chains per lane).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/int_rate tools/experiments/int_rate.hip && /tmp/int_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(NAME, EXPR)                                                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int n_outer, unsigned a, unsigned b)                        \
    {                                                                                                                      \
        unsigned x[8];                                                                                                     \
        _Pragma("unroll") for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 2654435761u + c;                                \
        for (int i = 0; i < n_outer; i++) {                                                                                \
            _Pragma("unroll") for (int k = 0; k < 30; k++) {                                                               \
                _Pragma("unroll") for (int c = 0; c < 8; c++) { unsigned v = x[c]; x[c] = (EXPR); }                        \
            }                                                                                                              \
        }                                                                                                                  \
        unsigned s = 0;                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 8; c++) s ^= x[c];                                                           \
        if (s == 0x12345678u) out[0] = s;                                                                                  \
    }

__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }

BODY(k_fma, f2u(__builtin_fmaf(u2f(v), u2f(a), u2f(b))))
__device__ __forceinline__ unsigned mul_lo_asm(unsigned v, unsigned a) { unsigned r; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ unsigned add_asm(unsigned v, unsigned a) { unsigned r; asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ unsigned mul24_asm(unsigned v, unsigned a) { unsigned r; asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ unsigned max_asm(unsigned v, unsigned a) { unsigned r; asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ unsigned xor_asm(unsigned v, unsigned a) { unsigned r; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
__device__ __forceinline__ unsigned mulhi_asm(unsigned v, unsigned a) { unsigned r; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(a)); return r; }
BODY(k_mul_lo, mul_lo_asm(v, a))
BODY(k_mul_hi, mulhi_asm(v, a))
BODY(k_mul_u24, mul24_asm(v, a))
BODY(k_xor_shift, xor_asm(v, a))
BODY(k_add, add_asm(v, a))
BODY(k_rcp, f2u(__builtin_amdgcn_rcpf(u2f(v))))
BODY(k_sqrt, f2u(__builtin_amdgcn_sqrtf(u2f(v))))
BODY(k_div_scale_like, f2u(__builtin_amdgcn_div_fixupf(u2f(v), u2f(a), u2f(b))))
BODY(k_cndmask, (v > a) ? v : b)
BODY(k_max, max_asm(v, a))

template <typename K>
static void run(const char* name, K kernel, unsigned* d, int num_cus, int per_iter)
{
    const int n_outer = 1 << 12;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kernel, dim3(num_cus * 4 * 4), dim3(256), 0, 0, d, n_outer, 0x7FEB352Du, 0x3f800001u);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double insts_per_wave = (double)n_outer * 240 * per_iter;
    const double wave_insts_per_simd = insts_per_wave * 4 * 4;  // 4 waves per SIMD x 4 rounds
    std::printf("%-18s %8.3f ms -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, best, best * 1e6 / wave_insts_per_simd * 2.4);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int num_cus = prop.multiProcessorCount;
    unsigned* d; hipMalloc(&d, 4);
    run("v_fma_f32", k_fma, d, num_cus, 1);
    run("v_mul_lo_u32", k_mul_lo, d, num_cus, 1);
    run("v_mul_u32_u24", k_mul_u24, d, num_cus, 1);
    run("v_xor_b32", k_xor_shift, d, num_cus, 1);
    run("v_mul_hi_u32", k_mul_hi, d, num_cus, 1);
    run("v_add_u32", k_add, d, num_cus, 1);
    run("v_rcp_f32", k_rcp, d, num_cus, 1);
    run("v_sqrt_f32", k_sqrt, d, num_cus, 1);
    run("v_div_fixup_f32", k_div_scale_like, d, num_cus, 1);
    run("v_cmp + v_cndmask (2)", k_cndmask, d, num_cus, 2);
    run("v_max_f32", k_max, d, num_cus, 1);
    return 0;
}
