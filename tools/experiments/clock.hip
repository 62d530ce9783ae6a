// clock.hip -- effective shader clock under a VALU-only load: one wave per SIMD runs a chain of N dependent v_fma_f32
// (4 cycles each on a 16-lane SIMD), and s_memtime / wall time give the frequency.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/clock tools/experiments/clock.hip && /tmp/clock
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void chain(float* out, int n, float a, float b)
{
    float x = threadIdx.x;
    for (int i = 0; i < n; i += 256) {
#pragma unroll
        for (int k = 0; k < 256; k++) x = __builtin_fmaf(x, a, b);
    }
    if (x == 12345.678f) out[0] = x;
}

int main()
{
    float* d; hipMalloc(&d, 4);
    const int n = 4 << 20;
    for (int waves_per_simd : { 1, 4 }) {
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(chain, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, n, 1.0000001f, 1e-9f);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2)
                std::printf("%d wave(s)/SIMD: %d dependent FMAs in %.3f ms -> %.2f GHz if 4 cycles per wave64 FMA per wave (x%d waves sharing the SIMD)\n",
                            waves_per_simd, n, ms, (double)n * 4 * waves_per_simd / (ms * 1e-3) / 1e9, waves_per_simd);
        }
    }
    return 0;
}
