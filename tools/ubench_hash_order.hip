// ubench_hash_order.hip — does the ORDER of the hash instructions inside one wave matter on gfx950?
//
// The headline kernel is issue-bound at 4.2 SIMD cycles per VALU instruction where the per-class costs of its mix would
// allow ~3.4 (profiles/pmc_valu.json); 75 % of its instructions are SHA-256(33-byte key) + RIPEMD-160(digest)
// (what the reference computes in src/shaders/sha256.wgsl:43-170 and src/shaders/ripemd160.wgsl:10-100).  This probe runs
// that hash pair — the same algorithm, the same constant folding of the padded message words — with every VALU
// instruction pinned in place by `asm volatile`, in several orders of the SAME instruction multiset, at 1..8 waves per SIMD:
//
//   compiler     the product's core/hash.h as hipcc schedules it (tools/ubench_hash.hip's loop: the 16.42 G pairs/s line)
//   asm_natural  one chain, instructions in dependency order as the round macros write them
//   asm_grouped  one chain, per round the half-rate instructions (v_alignbit, v_add3) first, then the full-rate ones
//                (v_bitop3, v_add) — "full-rate ops grouped behind each v_alignbit burst"
//   asm_x2       TWO independent keys per lane, interleaved instruction by instruction (two dependency chains per wave)
//   asm_x2_grouped  two keys, per round: both chains' half-rate instructions, then both chains' full-rate ones
//
// Reported per variant and occupancy: G hash pairs/s, shader MHz, SIMD cycles per wave-hash-pair and, with the VALU
// instruction count per pair read from the kernel's own ISA (tools/hash_order_census.py), cycles per instruction.
// Build: python tools/gen_hash_order.py   (writes tools/hash_order_gen.inc: generated, not kept in the repository)
//        hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_hash_order.hip -o tools/ubench_hash_order
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// pinned instructions (one asm volatile each: neither reordered among themselves nor folded)
__device__ __forceinline__ u32 i_alignbit(u32 hi, u32 lo, int s) {
    u32 r;
    asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "n"(s));
    return r;
}
template <int TT>
__device__ __forceinline__ u32 i_bitop3(u32 a, u32 b, u32 c) {
    u32 r;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(TT));
    return r;
}
// the same with one operand from the scalar file / a literal (what hipcc emits when an input is a compile-time constant)
template <int TT>
__device__ __forceinline__ u32 i_bitop3_s0(u32 a, u32 b, u32 c) {
    u32 r;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "s"(a), "v"(b), "v"(c), "n"(TT));
    return r;
}
template <int TT>
__device__ __forceinline__ u32 i_bitop3_s1(u32 a, u32 b, u32 c) {
    u32 r;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "s"(b), "v"(c), "n"(TT));
    return r;
}
template <int TT>
__device__ __forceinline__ u32 i_bitop3_s2(u32 a, u32 b, u32 c) {
    u32 r;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "v"(b), "s"(c), "n"(TT));
    return r;
}
__device__ __forceinline__ u32 i_add(u32 a, u32 b) {
    u32 r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 i_addk(u32 a, u32 k) {   // literal operand
    u32 r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(a));
    return r;
}
__device__ __forceinline__ u32 i_add3(u32 a, u32 b, u32 c) {
    u32 r;
    asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ u32 i_add3k(u32 a, u32 b, u32 k) {
    u32 r;
    asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
__device__ __forceinline__ u32 i_lshr(u32 a, int s) {
    u32 r;
    asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(r) : "n"(s), "v"(a));
    return r;
}

__device__ __forceinline__ u32 i_mov(u32 k) {
    u32 r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(k));
    return r;
}

// The hash pair as flat lists of the pinned instructions above, one function per order (tools/gen_hash_order.py executes
// the two compressions symbolically, folds what the padded message makes constant — as hipcc does in core/hash.h — and
// arranges the instructions): hashpair_natural / hashpair_grouped (one key), hashpair_x2 / hashpair_x2_grouped (two keys).
#include "hash_order_gen.inc"

template <int V>
__device__ __forceinline__ void hashpair(const u32 *prefix, const u32 (*xw)[8], u32 (*out)[5]) {
    if (V == 0) hashpair_natural(prefix, xw, out);
    else if (V == 1) hashpair_grouped(prefix, xw, out);
    else if (V == 2) hashpair_x2(prefix, xw, out);
    else hashpair_x2_grouped(prefix, xw, out);
}

// ---- kernels -----------------------------------------------------------------------------------------------------

// the product's code as hipcc schedules it (the loop of tools/ubench_hash.hip)
__global__ void __launch_bounds__(256) k_compiler(u32 *out, int iters, unsigned long long *clk) {
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
        xw[5] += h[0]; xw[6] += h[1]; xw[7] += h[2];
    }
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) r ^= h[i];
    if (r == 0x12345678u || iters < 0) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (iters == 1)   // check run: digest of the lane's first key
#pragma unroll
        for (int i = 0; i < 5; i++) out[(blockIdx.x * blockDim.x + threadIdx.x) * 5 + i] = h[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int NC, int V>
__global__ void __launch_bounds__(256) k_pinned(u32 *out, int iters, unsigned long long *clk) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    u32 xw[NC][8], h[NC][5], prefix[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int i = 0; i < 8; i++) xw[c][i] = threadIdx.x * 0x9E3779B9u + blockIdx.x * 0x85EBCA6Bu + i + c * 0x01000193u;
#pragma unroll
        for (int i = 0; i < 5; i++) h[c][i] = 0;
    }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < NC; c++) prefix[c] = 2u | (h[c][0] & 1u);
        hashpair<V>(prefix, xw, h);
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int i = 0; i < 5; i++) xw[c][i] ^= h[c][i];
            xw[c][5] += h[c][0]; xw[c][6] += h[c][1]; xw[c][7] += h[c][2];
        }
    }
    u32 r = 0;
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int i = 0; i < 5; i++) r ^= h[c][i];
    if (r == 0x12345678u || iters < 0) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (iters == 1)   // check run: digest of chain 0's first key (the same key k_compiler hashes)
#pragma unroll
        for (int i = 0; i < 5; i++) out[(blockIdx.x * blockDim.x + threadIdx.x) * 5 + i] = h[0][i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

typedef void (*kern_t)(u32 *, int, unsigned long long *);
struct Variant {
    const char *name;
    kern_t k;
    int chains;
};

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 256;
    const Variant vars[] = {{"compiler", k_compiler, 1},
                            {"asm_natural", k_pinned<1, 0>, 1},
                            {"asm_grouped", k_pinned<1, 1>, 1},
                            {"asm_x2", k_pinned<2, 2>, 2},
                            {"asm_x2_grouped", k_pinned<2, 3>, 2}};
    u32 *dout;
    unsigned long long *dclk;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const size_t out_words = (size_t)prop.multiProcessorCount * 8 * 256 * 5;
    CHECK(hipMalloc(&dout, out_words * sizeof(u32)));
    CHECK(hipMalloc(&dclk, 16));
    // correctness first: every variant's first digest equals the product code's, on every lane of one workgroup per CU
    {
        const int blocks = prop.multiProcessorCount;
        std::vector<u32> ref((size_t)blocks * 256 * 5), got(ref.size());
        hipLaunchKernelGGL(k_compiler, dim3(blocks), dim3(256), 0, 0, dout, 1, dclk);
        CHECK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
        for (const Variant &v : vars) {
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, dout, 1, dclk);
            CHECK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
            if (memcmp(ref.data(), got.data(), ref.size() * 4) != 0) {
                fprintf(stderr, "variant %s computes a different digest than core/hash.h\n", v.name);
                return 2;
            }
        }
        fprintf(stderr, "all %zu variants agree with core/hash.h on %d keys\n", sizeof vars / sizeof vars[0], blocks * 256);
    }
    for (const Variant &v : vars) {
        for (int w : {1, 2, 3, 4, 6, 8}) {
            const int blocks = prop.multiProcessorCount * w;   // 256 lanes = one wave on each of a CU's four SIMDs
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, dout, 8, dclk);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            double mhz = 0;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, dout, iters, dclk);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long clk[2];
                CHECK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
                if (ms < best) {
                    best = ms;
                    mhz = (double)clk[0] / (double)clk[1] * 100.0;
                }
            }
            const double pairs = (double)blocks * 256.0 * iters * v.chains;
            const double cyc = best * 1e-3 * mhz * 1e6 / ((double)w * iters * v.chains);   // SIMD cycles per wave-level hash pair
            printf("{\"variant\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.3f,\"Gpairs_per_s\":%.2f,\"shader_mhz\":%.0f,\"simd_cycles_per_wave_pair\":%.0f}\n",
                   v.name, w, best, pairs / (best * 1e-3) / 1e9, mhz, cyc);
            fflush(stdout);
        }
    }
    return 0;
}
