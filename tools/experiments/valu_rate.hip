// valu_rate.hip -- how many cycles one wave64 v_fma_f32 costs a gfx950 SIMD, as a function of the number of resident
// waves per SIMD and of whether a wave's instructions depend on each other.  Settles the rate used for the VALU roofline
// (profiles/r02_valu_rate.txt keeps the output; MI355X_MICROARCH.md: 2 cycles on the SIMD-32, 4 for one wave alone).
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/experiments/valu_rate.hip && /tmp/valu_rate
//
// Every workgroup is 256 threads = 4 waves = one wave per SIMD of its CU; the grid is 256 CUs x (waves per SIMD), and a
// 32 KB * (8 / waves-per-SIMD) LDS allocation per workgroup caps how many workgroups share a CU, so "w waves per SIMD" is
// what actually runs.  kChains independent accumulators per lane: 1 = every FMA depends on the previous one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int kChains>
__global__ __launch_bounds__(256) void fma_stream(float* out, int n_outer, float a, float b, unsigned long long* cycles)
{
    extern __shared__ float lds[];
    float x[kChains];
#pragma unroll
    for (int c = 0; c < kChains; c++) x[c] = (float)(threadIdx.x + c);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n_outer; i++) {
#pragma unroll
        for (int k = 0; k < 240 / kChains; k++) {
#pragma unroll
            for (int c = 0; c < kChains; c++) x[c] = __builtin_fmaf(x[c], a, b);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kChains; c++) s += x[c];
    if (s == 12345.678f) { out[0] = s; lds[threadIdx.x] = s; }
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int kChains>
static void run(int waves_per_simd, float* d, unsigned long long* d_cyc, int num_cus)
{
    const int n_outer = 1 << 13;                           // x 240 FMAs per lane
    constexpr int kRounds = 4;                             // the grid holds 4 rounds of full occupancy (placement need not be perfect)
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;  // at most `waves_per_simd` workgroups fit a CU
    hipFuncSetAttribute((const void*)fma_stream<kChains>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(fma_stream<kChains>, dim3(num_cus * waves_per_simd * kRounds), dim3(256), lds, 0, d, n_outer, 1.0000001f, 1e-9f, d_cyc);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
    const double fmas_per_wave = (double)n_outer * 240;  // 240 is divisible by every chain count used
    const double wave_insts_per_simd = fmas_per_wave * waves_per_simd * kRounds;
    // shader-clock cycles the SIMD spent per wave64 FMA it executed (s_memtime counts at a fixed 100 MHz, so the clock
    // is taken as 2.4 GHz nominal and, independently, the wall time is reported)
    std::printf("chains %2d  waves/SIMD %d : %8.3f ms  -> %.2f ns per wave-FMA per SIMD = %.2f cycles at 2.4 GHz  (one wave: %.2f cycles per FMA)  %.1f TFLOP/s\n",
                kChains, waves_per_simd, best, best * 1e6 / wave_insts_per_simd, best * 1e6 / wave_insts_per_simd * 2.4,
                best * 1e6 / (fmas_per_wave * kRounds) * 2.4, wave_insts_per_simd * 64 * 2 * num_cus * 4 / (best * 1e-3) / 1e12);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int num_cus = prop.multiProcessorCount;
    std::printf("%s, %d CUs, clockRate %d kHz\n", prop.name, num_cus, prop.clockRate);
    float* d; hipMalloc(&d, 4);
    unsigned long long* d_cyc; hipMalloc(&d_cyc, 8);
    // chains = distance (in instructions) between an FMA and the one it depends on; waves = resident waves per SIMD
    for (int w : { 1, 2, 3, 4, 8 }) run<1>(w, d, d_cyc, num_cus);
    for (int w : { 1, 2, 3, 4, 8 }) run<2>(w, d, d_cyc, num_cus);
    for (int w : { 1, 2, 3, 4, 8 }) run<3>(w, d, d_cyc, num_cus);
    for (int w : { 1, 2, 3, 4, 8 }) run<4>(w, d, d_cyc, num_cus);
    for (int w : { 1, 2, 4 }) run<6>(w, d, d_cyc, num_cus);
    for (int w : { 1, 2, 4, 8 }) run<8>(w, d, d_cyc, num_cus);
    return 0;
}
