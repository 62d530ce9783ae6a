// apicost.hip -- host cost of the HIP calls a frame is made of (tools/experiments/README.md).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/apicost tools/experiments/apicost.hip && /tmp/apicost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { char b[420]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_big(Big a, int* p) { if (p && threadIdx.x == 12345) *p = a.b[3]; }

template <typename F>
double per_call_us(int n, F&& f)
{
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f(i);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

int main()
{
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[8];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t evt[2];
    for (auto& e : evt) hipEventCreate(&e);
    int* d; hipMalloc(&d, 1 << 20);
    int* h; hipHostMalloc(&h, 1 << 20);
    Big big{};
    const int N = 2000;
    for (int rep = 0; rep < 2; rep++) {
        double a = per_call_us(N, [&](int) { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s[0], d); });
        hipDeviceSynchronize();
        double b = per_call_us(N, [&](int) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s[0], big, d); });
        hipDeviceSynchronize();
        double b4 = per_call_us(N, [&](int i) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s[i & 3], big, d); });
        hipDeviceSynchronize();
        double c = per_call_us(N, [&](int i) { hipEventRecord(ev[i & 7], s[0]); });
        hipDeviceSynchronize();
        double ct = per_call_us(N, [&](int i) { hipEventRecord(evt[i & 1], s[0]); });
        hipDeviceSynchronize();
        double e = per_call_us(N, [&](int i) { hipEventRecord(ev[i & 7], s[0]); hipStreamWaitEvent(s[1], ev[i & 7], 0); });
        hipDeviceSynchronize();
        double f = per_call_us(N, [&](int) { hipMemcpyAsync(h, d, 256, hipMemcpyDeviceToHost, s[0]); });
        hipDeviceSynchronize();
        double g = per_call_us(N, [&](int) { hipMemcpyAsync(d, h, 256, hipMemcpyHostToDevice, s[0]); });
        hipDeviceSynchronize();
        double kk = per_call_us(N, [&](int) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s[0], big, d); hipMemcpyAsync(h, d, 256, hipMemcpyDeviceToHost, s[0]); });
        hipDeviceSynchronize();
        if (rep)
            std::printf("launch(small args) %.2f us | launch(420 B args) %.2f us | same over 4 streams %.2f us | eventRecord(no timing) %.2f us | eventRecord(timing) %.2f us | "
                        "record+streamWait %.2f us | memcpyAsync D2H 256 B %.2f us | H2D %.2f us | launch+D2H %.2f us\n", a, b, b4, c, ct, e, f, g, kk);
    }
    // a 3-kernel graph replayed (kernel params fixed): what hipGraphLaunch costs on this runtime
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s[0], big, d);
    hipStreamEndCapture(s[0], &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    for (int rep = 0; rep < 2; rep++) {
        double g = per_call_us(N, [&](int) { hipGraphLaunch(exec, s[0]); });
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) hipGraphLaunch(exec, s[0]);
        hipDeviceSynchronize();
        double tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) for (int j = 0; j < 3; j++) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s[0], big, d);
        hipDeviceSynchronize();
        double tot2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / N;
        if (rep) std::printf("graph of 3 kernels: hipGraphLaunch %.2f us host, %.2f us end-to-end per replay; 3 plain launches end-to-end %.2f us\n", g, tot, tot2);
    }
    return 0;
}
