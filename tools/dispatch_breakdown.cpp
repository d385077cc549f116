// dispatch_breakdown.cpp — host-side cost of the HIP calls one vgen_dispatch is made of, measured in isolation on an
// idle stream: kernel launch with a 2.4 KB argument block vs a small one, event record, stream-wait-event, async copy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { unsigned w[600]; };
__global__ void kbig(const Big a, unsigned *out) { if (out && threadIdx.x == 12345) out[0] = a.w[5]; }
__global__ void ksmall(unsigned *out, unsigned v) { if (out && threadIdx.x == 12345) out[0] = v; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e1, e2; CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    unsigned *d, *h; CK(hipMalloc(&d, 1 << 16)); CK(hipHostMalloc(&h, 1 << 16));
    Big b{}; const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [&](auto a) { return std::chrono::duration<double, std::micro>(now() - a).count() / N; };
    for (int rep = 0; rep < 2; rep++) {
        auto t = now(); for (int i = 0; i < N; i++) hipLaunchKernelGGL(kbig, dim3(1), dim3(64), 0, s, b, d); double a = us(t); CK(hipStreamSynchronize(s));
        t = now(); for (int i = 0; i < N; i++) hipLaunchKernelGGL(ksmall, dim3(1), dim3(64), 0, s, d, 1u); double c = us(t); CK(hipStreamSynchronize(s));
        t = now(); for (int i = 0; i < N; i++) CK(hipEventRecord(e1, s)); double er = us(t); CK(hipStreamSynchronize(s));
        t = now(); for (int i = 0; i < N; i++) CK(hipEventRecord(e2, s)); double er2 = us(t); CK(hipStreamSynchronize(s));
        t = now(); for (int i = 0; i < N; i++) CK(hipMemcpyAsync(h, d, 10256, hipMemcpyDeviceToHost, s)); double mc = us(t); CK(hipStreamSynchronize(s));
        t = now(); for (int i = 0; i < N; i++) { hipLaunchKernelGGL(ksmall, dim3(1), dim3(64), 0, s, d, 1u); CK(hipEventRecord(e2, s)); CK(hipEventSynchronize(e2)); } double rt = us(t);
        t = now(); for (int i = 0; i < N; i++) { hipLaunchKernelGGL(ksmall, dim3(1), dim3(64), 0, s, d, 1u); CK(hipStreamSynchronize(s)); } double rt2 = us(t);
        printf("launch 2.4KB args %.2f us | launch small %.2f | eventRecord(timing) %.2f | eventRecord(no timing) %.2f | memcpyAsync D2H 10KB %.2f | launch+record+eventSync round trip %.2f | launch+streamSync %.2f\n", a, c, er, er2, mc, rt, rt2);
    }
    return 0;
}
