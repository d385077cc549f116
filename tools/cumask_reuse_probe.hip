// cumask_reuse_probe.hip — does creating CU-masked streams after destroying earlier ones work on this runtime?
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void nop(unsigned *p) { if (p && threadIdx.x == 0) atomicAdd(p, 1u); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int words = (prop.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
    unsigned *d;
    CK(hipMalloc(&d, 1 << 20));
    for (int round = 0; round < 4; round++) {
        const int k = 4 << (round & 1);
        std::vector<hipStream_t> st(k);
        for (auto &s : st) {
            CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
            printf("round %d: created\n", round); fflush(stdout);
        }
        CK(hipMemsetAsync(d, 0, 1 << 20, st[0]));
        printf("round %d: memset enqueued\n", round); fflush(stdout);
        CK(hipStreamSynchronize(st[0]));
        for (auto s : st) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, d);
        for (auto s : st) CK(hipStreamSynchronize(s));
        printf("round %d: %d streams launched + synced\n", round, k); fflush(stdout);
        for (auto s : st) CK(hipStreamDestroy(s));
        printf("round %d: destroyed\n", round); fflush(stdout);
    }
    printf("ok\n");
    return 0;
}
