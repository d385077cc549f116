// ubench_hash_yield.hip — do issue-slot yields in the hash pair pay when the waves of a SIMD are OUT OF STEP?
//
// In the real kernels the hash pair as a generated block with one `s_nop 0` per three instructions (device/hashgen.py) gained
// 5-6 % — but only with launches of several frames resident, never for one launch alone (profiles/r04_hash_blocks_ab.txt).
// rocprofv3 serialises kernels under --pmc, so no counter pass sees that state.  This probe isolates it: ONE launch of the hash
// pair alone (no point arithmetic), four waves per SIMD, in two regimes —
//   lockstep   all waves start together (what one launch of the scan does),
//   staggered  every workgroup first sleeps 0..3 quarters of a hash pair's duration (by a hash of its index), so that the four
//              waves of a SIMD are at different places of the instruction list (what waves of different launches are) —
// for  compiler  hipcc's schedule of core/hash.h  and six generated blocks (default: no yields, a yield per 3 / 2 / 4 instructions, ...).
//
// Build: bash tools/ubench_hash_yield_gen.sh [six hashgen.py --yield modes]     (writes tools/hb_*.inc, tools/hb_labels.h: generated, not
//        kept in the repository; then compiles tools/ubench_hash_yield)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../vgen_amd/csrc/core/hash.h"

using namespace vg;

// tools/ubench_hash_yield_gen.sh writes hb_0.inc .. hb_5.inc (one hashgen.py --yield mode each) and hb_labels.h (their names)
namespace hb0 {
#include "hb_0.inc"
}
namespace hb1 {
#include "hb_1.inc"
}
namespace hb2 {
#include "hb_2.inc"
}
namespace hb3 {
#include "hb_3.inc"
}
namespace hb4 {
#include "hb_4.inc"
}
namespace hb5 {
#include "hb_5.inc"
}
#include "hb_labels.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

// HB_CALL_i(prefix, xw, h, fill): hb_labels.h - the variant's block, with the filler's sink where the variant carries multiply-adds
template <int V>
__device__ __forceinline__ void hashpair(u32 prefix, const u32 xw[8], u32 h[5], u32 *fill) {
    if (V == 0) {
        u32 sha[8];
        sha256_pub33(prefix, xw, sha);
        ripemd160_of_sha(sha, h);
    } else if (V == 1) {
        HB_CALL_0(prefix, xw, h, fill);
    } else if (V == 2) {
        HB_CALL_1(prefix, xw, h, fill);
    } else if (V == 3) {
        HB_CALL_2(prefix, xw, h, fill);
    } else if (V == 4) {
        HB_CALL_3(prefix, xw, h, fill);
    } else if (V == 5) {
        HB_CALL_4(prefix, xw, h, fill);
    } else {
        HB_CALL_5(prefix, xw, h, fill);
    }
}

// quarters: 0 = lockstep; q > 0: workgroup b sleeps (mix(b) & 3) * q times s_sleep(127) (8 128 cycles each) before it starts
template <int V>
__global__ void __launch_bounds__(256) k_hash(u32 *out, int iters, int quarters, unsigned long long *clk) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    const u32 phase = ((blockIdx.x * 2654435761u) >> 13) & 3u;
    for (u32 i = 0; i < phase * (u32)quarters; i++) __builtin_amdgcn_s_sleep(127);
    u32 xw[8], h[5] = {0, 0, 0, 0, 0}, fill[1] = {0};
#pragma unroll
    for (int i = 0; i < 8; i++) xw[i] = threadIdx.x * 0x9E3779B9u + blockIdx.x * 0x85EBCA6Bu + i;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        hashpair<V>(2u | (h[0] & 1u), xw, h, fill);
#pragma unroll
        for (int i = 0; i < 5; i++) xw[i] ^= h[i];
        xw[5] += h[0]; xw[6] += h[1]; xw[7] += h[2];
    }
    u32 r = fill[0];
#pragma unroll
    for (int i = 0; i < 5; i++) r ^= h[i];
    if (r == 0x12345678u || iters < 0) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (iters == 1)
#pragma unroll
        for (int i = 0; i < 5; i++) out[(blockIdx.x * blockDim.x + threadIdx.x) * 5 + i] = h[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

typedef void (*kern_t)(u32 *, int, int, unsigned long long *);

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 512;
    const struct { const char *name; kern_t k; } vars[] = {{"compiler", k_hash<0>}, {HB_LABEL_0, k_hash<1>}, {HB_LABEL_1, k_hash<2>}, {HB_LABEL_2, k_hash<3>},
                                                           {HB_LABEL_3, k_hash<4>}, {HB_LABEL_4, k_hash<5>}, {HB_LABEL_5, k_hash<6>}};
    u32 *dout;
    unsigned long long *dclk;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    CHECK(hipMalloc(&dout, (size_t)prop.multiProcessorCount * 8 * 256 * 5 * sizeof(u32)));
    CHECK(hipMalloc(&dclk, 16));
    {
        const int blocks = prop.multiProcessorCount;
        std::vector<u32> ref((size_t)blocks * 256 * 5), got(ref.size());
        hipLaunchKernelGGL(vars[0].k, dim3(blocks), dim3(256), 0, 0, dout, 1, 0, dclk);
        CHECK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
        for (const auto &v : vars) {
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, dout, 1, 0, dclk);
            CHECK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
            if (memcmp(ref.data(), got.data(), ref.size() * 4) != 0) {
                fprintf(stderr, "variant %s computes a different digest than core/hash.h\n", v.name);
                return 2;
            }
        }
        fprintf(stderr, "all variants agree with core/hash.h on %d keys\n", blocks * 256);
    }
    for (int w : {4, 2, 8}) {
        for (int quarters : {0, 2}) {
            for (const auto &v : vars) {
                const int blocks = prop.multiProcessorCount * w;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0));
                CHECK(hipEventCreate(&e1));
                hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, dout, 8, quarters, dclk);
                CHECK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, dout, iters, quarters, dclk);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                // the sleeps are part of the launch: at most 3 * quarters * 8 128 cycles (~10-20 us) against milliseconds of hashing
                const double pairs = (double)blocks * 256.0 * iters;
                printf("{\"variant\":\"%s\",\"waves_per_simd\":%d,\"stagger_quarters\":%d,\"ms\":%.3f,\"Gpairs_per_s\":%.2f}\n", v.name, w, quarters, best,
                       pairs / (best * 1e-3) / 1e9);
                fflush(stdout);
            }
        }
    }
    return 0;
}
