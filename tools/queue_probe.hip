// queue_probe.hip — how many kernels of DIFFERENT streams does the device run at once, by stream kind?
// K streams, one 200 us one-wave sleeper each: time = 200 us x rounds.  Kinds: ordinary (normal priority),
// priorities cycled (greatest..least), CU-masked (all CUs enabled).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void sleeper(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const unsigned long long ticks = 20000;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int words = (prop.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int kind = 0; kind < 3; kind++) {
        std::vector<hipStream_t> st;
        for (int k = 1; k <= 16; k++) {
            hipStream_t s;
            if (kind == 0) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            else if (kind == 1) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi + (k - 1) % (lo - hi + 1)));
            else CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
            st.push_back(s);
            hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, s, 100ull);   // warm the stream (queue assignment)
            CK(hipStreamSynchronize(s));
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                auto t0 = now();
                for (auto x : st) hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, x, ticks);
                for (auto x : st) CK(hipStreamSynchronize(x));
                best = std::min(best, std::chrono::duration<double, std::micro>(now() - t0).count());
            }
            printf("kind=%s streams=%2d: %7.1f us (%.1f rounds)\n", kind == 0 ? "ordinary" : kind == 1 ? "priority" : "cumask  ", k, best, best / 200.0);
        }
        for (auto x : st) CK(hipStreamDestroy(x));
    }
    return 0;
}
