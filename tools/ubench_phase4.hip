// ubench_phase4.hip — which place of an issue slot does an opcode take?  Per opcode: alone (four waves); as the OLDEST wave beside three waves of
// v_add_u32 (does a full-rate instruction ride behind it?); as the three YOUNGER waves beside one wave of v_alignbit_b32 (does it ride itself?).
//
// Follow-up of tools/ubench_phase.hip (profiles/r05_phase_ubench.jsonl: with the waves of a SIMD barrier-locked, v_add_u32 alone issues at 2.0
// cycles, v_alignbit_b32 and v_mad_u64_u32 alone at 4.0, and every mix of add and alignbit in runs of up to 64 at 3.9 - 4.0 PER INSTRUCTION: beside
// half-rate instructions the full-rate ones cost four cycles as well).  Here:
//   (1) which SIMD the waves of a 256 / 512 / 1024-thread workgroup land on (s_getreg HW_ID);
//   (2) uniform streams with a class ratio other than 1:1 (7:1, 3:1, 1:3, 15:1 ... of full-rate : half-rate), every wave the same;
//   (3) waves SPECIALISED by class: of the four waves of a SIMD, the `split` lowest run one stream, the others another.
// Time is taken per block of 1 024 instructions per wave between workgroup barriers; reported as SIMD cycles per block and per wave-instruction.
// Build: python3 tools/ubench_phase2_gen.py tools/ubench_phase4_blocks.inc && hipcc --offload-arch=gfx950 -O3 tools/ubench_phase2.hip -o tools/ubench_phase2
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int CHAINS = 8;
constexpr int BLOCK_INSTR = 1024;

#include "ubench_phase4_blocks.inc"

#define RUN_BLOCK(S)                                                                                                                  \
    asm volatile(S : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(acc[0]),   \
                 "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])                     \
                 : "v"(y), "v"(z)                                                                                                     \
                 : "vcc")

__device__ __forceinline__ uint32_t hw_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

__global__ void k_map(uint32_t *out) {
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw_id();
}

// rank of this wave among the waves of its workgroup that sit on the same SIMD (by wave index)
__device__ __forceinline__ uint32_t simd_rank(uint32_t *lds) {
    const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t simd = (hw_id() >> 4) & 3u;
    if ((threadIdx.x & 63) == 0) lds[wave] = simd;
    __syncthreads();
    uint32_t r = 0;
    for (uint32_t v = 0; v < nw; v++) r += (v < wave && lds[v] == simd) ? 1u : 0u;
    return __builtin_amdgcn_readfirstlane(r);
}

#define KERNEL_PROLOGUE                                                                       \
    __shared__ uint32_t lds[16];                                                              \
    uint32_t x[CHAINS], y = threadIdx.x | 1u, z = threadIdx.x * 2654435761u + 12345u;         \
    uint64_t acc[CHAINS];                                                                     \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) {                                      \
        x[c] = threadIdx.x * 747796405u + c * 2891336453u + blockIdx.x;                       \
        acc[c] = ((uint64_t)x[c] << 32) | (x[c] ^ 0x9E3779B9u);                               \
    }                                                                                         \
    const uint32_t rank = simd_rank(lds);                                                     \
    __syncthreads();                                                                          \
    const unsigned long long c0 = clock64();

#define KERNEL_EPILOGUE                                                                                       \
    __syncthreads();                                                                                          \
    const unsigned long long c1 = clock64();                                                                  \
    uint32_t r = rank;                                                                                        \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) r ^= x[c] ^ (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32); \
    if (r == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                     \
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;

// every wave the same stream
#define UNIFORM(NAME, S)                                                                                        \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, int split, unsigned long long *clk) { \
        KERNEL_PROLOGUE                                                                                         \
        _Pragma("unroll 1") for (int it = 0; it < iters; it++) {                                                \
            __builtin_amdgcn_s_barrier();                                                                       \
            RUN_BLOCK(S);                                                                                       \
        }                                                                                                       \
        KERNEL_EPILOGUE                                                                                         \
    }

// the `split` lowest-ranked waves of every SIMD run S0, the others S1
#define ROLES(NAME, S0, S1)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, int iters, int split, unsigned long long *clk) { \
        KERNEL_PROLOGUE                                                                                         \
        if ((int)rank < split) {                                                                                \
            _Pragma("unroll 1") for (int it = 0; it < iters; it++) {                                            \
                __builtin_amdgcn_s_barrier();                                                                   \
                RUN_BLOCK(S0);                                                                                  \
            }                                                                                                   \
        } else {                                                                                                \
            _Pragma("unroll 1") for (int it = 0; it < iters; it++) {                                            \
                __builtin_amdgcn_s_barrier();                                                                   \
                RUN_BLOCK(S1);                                                                                  \
            }                                                                                                   \
        }                                                                                                       \
        KERNEL_EPILOGUE                                                                                         \
    }

#define DEFS(K) UNIFORM(u_##K, BLK_##K) ROLES(o_##K, BLK_##K, BLK_add) ROLES(y_##K, BLK_alignbit, BLK_##K)
ALL_OPS(DEFS)
UNIFORM(u_add, BLK_add)
UNIFORM(u_alignbit, BLK_alignbit)

typedef void (*kern_t)(uint32_t *, int, int, unsigned long long *);

static double run(kern_t k, int split, int iters, uint32_t *dout, unsigned long long *dclk, int cus) {
    hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, dout, 4, split, dclk);
    CHECK(hipDeviceSynchronize());
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 0, 0, dout, iters, split, dclk);
        CHECK(hipDeviceSynchronize());
        unsigned long long c;
        CHECK(hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost));
        if (c < best) best = c;
    }
    return (double)best / iters;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *dout;
    unsigned long long *dclk;
    CHECK(hipMalloc(&dout, (size_t)cus * 1024 * sizeof(uint32_t)));
    CHECK(hipMalloc(&dclk, 16));
    const double add = run(u_add, 4, iters, dout, dclk, cus), rot = run(u_alignbit, 4, iters, dout, dclk, cus);
    printf("{\"op\":\"add\",\"alone_cycles_per_instr\":%.3f}\n{\"op\":\"alignbit\",\"alone_cycles_per_instr\":%.3f}\n", add / 4096, rot / 4096);
    // cycles per block of 4 x 1 024 instructions: alone | one wave of the opcode (oldest) + three of add | one wave of alignbit (oldest) + three of the opcode
#define RUN(K)                                                                                                                           \
    {                                                                                                                                    \
        const double a = run(u_##K, 4, iters, dout, dclk, cus), o = run(o_##K, 1, iters, dout, dclk, cus), y = run(y_##K, 1, iters, dout, dclk, cus); \
        printf("{\"op\":\"%s\",\"alone_cycles_per_instr\":%.3f,\"oldest_beside_3_add_cycles_per_block\":%.0f,\"younger_beside_alignbit_cycles_per_block\":%.0f}\n", #K, a / 4096, o, y); \
        fflush(stdout);                                                                                                                  \
    }
    ALL_OPS(RUN)
    return 0;
}
