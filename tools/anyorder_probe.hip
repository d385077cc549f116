// anyorder_probe.hip — does a HIP stream on this box overlap consecutive kernels when they are launched with
// hipExtAnyOrderLaunch (AQL barrier bit cleared)?  Eight launches of a one-wave kernel that sleeps `us`
// microseconds: serialised they take 8*us, overlapped ~us.  Prints one line per launch pattern.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void sleeper(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (sink && threadIdx.x == 0) atomicAdd(sink, 1u);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int K = 8;
    const unsigned long long ticks = 20000;   // 200 us of the 100 MHz counter
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    std::vector<hipStream_t> st(K);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev[K];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    // warm
    hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[0], 100ull, sink);
    CK(hipStreamSynchronize(st[0]));
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[0], ticks, sink);
        CK(hipStreamSynchronize(st[0]));
        printf("same stream, ordinary launches:        %8.1f us (serial = %d)\n", us(t0, now()), (int)(K * ticks / 100));
        t0 = now();
        for (int i = 0; i < K; i++)
            hipExtLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[0], nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
        CK(hipStreamSynchronize(st[0]));
        printf("same stream, hipExtAnyOrderLaunch:     %8.1f us\n", us(t0, now()));
        t0 = now();
        for (int i = 0; i < K; i++) {
            hipExtLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[0], nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
            CK(hipEventRecord(ev[i], st[0]));
        }
        CK(hipStreamSynchronize(st[0]));
        printf("same stream, any-order + event record: %8.1f us\n", us(t0, now()));
        t0 = now();
        for (int i = 0; i < K; i++) {
            CK(hipStreamWaitEvent(st[0], ev[i], 0));   // already complete events
            hipExtLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[0], nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
        }
        CK(hipStreamSynchronize(st[0]));
        printf("same stream, wait(done event)+any-order:%7.1f us\n", us(t0, now()));
        t0 = now();
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[i], ticks, sink);
        for (int i = 0; i < K; i++) CK(hipStreamSynchronize(st[i]));
        printf("%d streams, ordinary launches:           %8.1f us\n", K, us(t0, now()));
        t0 = now();
        for (int i = 0; i < 4; i++) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[i], ticks, sink);
        for (int i = 0; i < 4; i++) CK(hipStreamSynchronize(st[i]));
        printf("4 streams, ordinary launches:           %8.1f us (1 each)\n", us(t0, now()));
        // kernel timing through the launch's own start/stop events (no extra packets?)
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        hipExtLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, st[0], a, b, 0, ticks, sink);
        CK(hipStreamSynchronize(st[0]));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("hipExtLaunchKernel start/stop events:  %8.1f us\n", ms * 1e3);
    }
    // do streams of different priorities / CU-masked streams get hardware queues of their own, beyond GPU_MAX_HW_QUEUES?
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d .. greatest %d\n", lo, hi);
    {
        std::vector<hipStream_t> ps;
        for (int p = hi; p <= lo; p++)
            for (int i = 0; i < 4; i++) {
                hipStream_t s;
                CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p));
                ps.push_back(s);
            }
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = now();
            for (auto s : ps) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, s, ticks, sink);
            for (auto s : ps) CK(hipStreamSynchronize(s));
            printf("%zu streams (4 per priority level), one launch each: %8.1f us\n", ps.size(), us(t0, now()));
        }
    }
    {
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        const int words = (prop.multiProcessorCount + 31) / 32;
        std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
        std::vector<hipStream_t> ms(12);
        for (auto &s : ms) CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = now();
            for (auto s : ms) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, s, ticks, sink);
            for (auto s : ms) CK(hipStreamSynchronize(s));
            printf("%zu CU-masked streams (all CUs), one launch each:      %8.1f us\n", ms.size(), us(t0, now()));
        }
    }
    return 0;
}
