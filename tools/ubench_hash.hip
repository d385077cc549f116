// ubench_hash.hip — how fast does gfx950 issue the path's real hash code when nothing else is in the way?
// Every lane runs SHA-256(33-byte compressed key) -> RIPEMD-160 `iters` times on register data (the output is
// fed back into the input), at 1/2/4 waves per SIMD.  Compared with the per-class issue-cost model
// (tools/isa_census.py) it separates "the model is optimistic for this mix" from "the scan kernels lose cycles
// elsewhere".  Build: hipcc --offload-arch=gfx950 -O3 -I. tools/ubench_hash.hip -o tools/ubench_hash
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../vgen_amd/csrc/core/hash.h"

using namespace vg;

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

__global__ void __launch_bounds__(256) hash_loop_kernel(u32 *out, int iters, unsigned long long *clk) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    u32 xw[8], h[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) xw[i] = threadIdx.x * 0x9E3779B9u + blockIdx.x * 0x85EBCA6Bu + i;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        u32 sha[8];
        sha256_pub33(2u | (h[0] & 1u), xw, sha);
        ripemd160_of_sha(sha, h);
#pragma unroll
        for (int i = 0; i < 5; i++) xw[i] ^= h[i];
        xw[5] += h[0];
        xw[6] += h[1];
        xw[7] += h[2];
    }
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) r ^= h[i];
    if (r == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 256;
    u32 *dout;
    unsigned long long *dclk;
    CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(u32)));
    CHECK(hipMalloc(&dclk, 16));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    for (int w : {1, 2, 3, 4, 8}) {
        const int blocks = prop.multiProcessorCount * w;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(hash_loop_kernel, dim3(blocks), dim3(256), 0, 0, dout, 8, dclk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(hash_loop_kernel, dim3(blocks), dim3(256), 0, 0, dout, iters, dclk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk[2];
        CHECK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
        const double mhz = (double)clk[0] / (double)clk[1] * 100.0;
        const double keys = (double)blocks * 256.0 * iters;
        const double cyc_per_wave_key = ms * 1e-3 * mhz * 1e6 / ((double)w * iters);   // SIMD cycles per 64 keys
        printf("{\"waves_per_simd\":%d,\"ms\":%.3f,\"Ghash_per_s\":%.2f,\"shader_mhz\":%.0f,\"simd_cycles_per_wave_hash\":%.0f}\n", w, ms,
               keys / (ms * 1e-3) / 1e9, mhz, cyc_per_wave_key);
    }
    return 0;
}
