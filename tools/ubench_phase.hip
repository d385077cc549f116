// ubench_phase.hip — does the PHASE between the waves of a SIMD matter for a mixed full-rate / half-rate integer stream?
//
// profiles/r01_ubench_valu.jsonl: v_add_u32 alone issues at 2.3 cycles per wave-instruction (four waves per SIMD), v_alignbit_b32 alone at
// 4.35, but any 1:1 mix of the two at 3.85 instead of the 3.33 their sum predicts — whatever the run length INSIDE a wave (1, 8, 32).  With
// four waves at arbitrary phases the SIMD sees a random interleave of the classes either way, so that measurement cannot tell a cost per
// CLASS SWITCH at the SIMD from anything else.  This probe can: the waves of a SIMD come from ONE workgroup (1024 threads: waves w, w+4, w+8, w+12
// share SIMD w % 4; 512 threads: two per SIMD), meet at an s_barrier before every block of ~2 000 instructions, and then run
//   inphase    every wave  [R x add, R x alignbit] ...          (the SIMD sees runs of one class from all its waves)
//   antiphase  half of the SIMD's waves run [R x alignbit, R x add] ...   (the SIMD sees both classes all the time)
// If inphase is clearly faster, phase-locking the waves of a SIMD (one workgroup, a barrier per key) is a lever for the hash pair;
// if not, the mixing cost is intrinsic to the stream and there is nothing to align.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_phase.hip -o tools/ubench_phase
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int CHAINS = 8;
constexpr int BLOCK_INSTR = 2048;   // VALU instructions between two barriers (a hash pair is 2 196)

#include "ubench_phase_blocks.inc"   // tools/ubench_phase_gen.py: BLK_<R>_<P>_<Q>_Y<n>, one asm statement of 2 048 instructions each

#define RUN_BLOCK(S)                                                                                                                  \
    asm volatile(S : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(acc[0]),   \
                 "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])                     \
                 : "v"(y), "v"(z)                                                                                                     \
                 : "vcc")

// MODE 0: inphase (all waves P,Q)   1: antiphase (waves with bit 2 of the wave index set run Q,P)   2 / 3: the same without barriers
#define DEFINE_KERNEL(NAME, WG, SPQ, SQP)                                                                                  \
    __global__ void __launch_bounds__(WG) NAME(uint32_t *out, int iters, int mode, unsigned long long *clk) {             \
        const unsigned long long c0 = clock64();                                                                           \
        uint32_t x[CHAINS], y = threadIdx.x | 1u, z = threadIdx.x * 2654435761u + 12345u;                                   \
        uint64_t acc[CHAINS];                                                                                              \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) {                                                               \
            x[c] = threadIdx.x * 747796405u + c * 2891336453u + blockIdx.x;                                                \
            acc[c] = ((uint64_t)x[c] << 32) | (x[c] ^ 0x9E3779B9u);                                                        \
        }                                                                                                                  \
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                            \
        const bool swap = (mode & 1) && ((wave >> 2) & 1u); /* waves w and w + 4 share a SIMD */                           \
        const bool bar = mode < 2;                                                                                         \
        if (swap) {                                                                                                        \
            _Pragma("unroll 1") for (int it = 0; it < iters; it++) {                                                       \
                if (bar) __builtin_amdgcn_s_barrier();                                                                     \
                RUN_BLOCK(SQP);                                                                                            \
            }                                                                                                              \
        } else {                                                                                                           \
            _Pragma("unroll 1") for (int it = 0; it < iters; it++) {                                                       \
                if (bar) __builtin_amdgcn_s_barrier();                                                                     \
                RUN_BLOCK(SPQ);                                                                                            \
            }                                                                                                              \
        }                                                                                                                  \
        uint32_t r = 0;                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) r ^= x[c] ^ (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32);         \
        if (r == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0;                                                  \
    }

typedef void (*kern_t)(uint32_t *, int, int, unsigned long long *);

static void run(const char *name, int R, int yield_every, int WG, kern_t k, int iters, uint32_t *dout, unsigned long long *dclk, int cus) {
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(k, dim3(cus), dim3(WG), 0, 0, dout, 4, mode, dclk);
        CHECK(hipDeviceSynchronize());
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k, dim3(cus), dim3(WG), 0, 0, dout, iters, mode, dclk);
            CHECK(hipDeviceSynchronize());
            unsigned long long c;
            CHECK(hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost));
            if (c < best) best = c;
        }
        // SIMD cycles per wave-instruction: the kernel's cycles / (VALU instructions per wave x waves per SIMD)
        const double per = (double)best / ((double)iters * BLOCK_INSTR * (WG / 256));
        printf("{\"mix\":\"%s\",\"run\":%d,\"yield_every\":%d,\"waves_per_simd\":%d,\"mode\":\"%s\",\"cycles_per_waveinstr\":%.3f}\n", name, R, yield_every,
               WG / 256, mode == 0 ? "inphase+barrier" : mode == 1 ? "antiphase+barrier" : mode == 2 ? "inphase,free" : "antiphase,free", per);
        fflush(stdout);
    }
}

#define CASE(R, P, Q, Y, WG) DEFINE_KERNEL(k_##R##_##P##_##Q##_y##Y##_##WG, WG, BLK_##R##_##P##_##Q##_Y##Y, BLK_##R##_##Q##_##P##_Y##Y)
CASE(8, 0, 0, 0, 1024)
CASE(8, 1, 1, 0, 1024)
CASE(8, 2, 2, 0, 1024)
CASE(1, 0, 1, 0, 1024)
CASE(4, 0, 1, 0, 1024)
CASE(16, 0, 1, 0, 1024)
CASE(64, 0, 1, 0, 1024)
CASE(256, 0, 1, 0, 1024)
CASE(1024, 0, 1, 0, 1024)
CASE(16, 3, 1, 0, 1024)
CASE(16, 0, 2, 0, 1024)
CASE(256, 0, 2, 0, 1024)
CASE(16, 1, 2, 0, 1024)
CASE(8, 0, 0, 3, 1024)
CASE(8, 1, 1, 3, 1024)
CASE(1, 0, 1, 3, 1024)
CASE(16, 0, 1, 3, 1024)
CASE(256, 0, 1, 3, 1024)
CASE(1024, 0, 1, 3, 1024)
CASE(16, 0, 1, 0, 512)
CASE(256, 0, 1, 0, 512)
CASE(1024, 0, 1, 0, 512)

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *dout;
    unsigned long long *dclk;
    CHECK(hipMalloc(&dout, (size_t)cus * 1024 * sizeof(uint32_t)));
    CHECK(hipMalloc(&dclk, 16));
#define RUN(NAME, R, P, Q, Y, WG) run(NAME, R, Y, WG, k_##R##_##P##_##Q##_y##Y##_##WG, iters, dout, dclk, cus)
    RUN("add only", 8, 0, 0, 0, 1024);
    RUN("alignbit only", 8, 1, 1, 0, 1024);
    RUN("mad64 only", 8, 2, 2, 0, 1024);
    RUN("add/alignbit", 1, 0, 1, 0, 1024);
    RUN("add/alignbit", 4, 0, 1, 0, 1024);
    RUN("add/alignbit", 16, 0, 1, 0, 1024);
    RUN("add/alignbit", 64, 0, 1, 0, 1024);
    RUN("add/alignbit", 256, 0, 1, 0, 1024);
    RUN("add/alignbit", 1024, 0, 1, 0, 1024);
    RUN("bitop3/alignbit", 16, 3, 1, 0, 1024);
    RUN("add/mad64", 16, 0, 2, 0, 1024);
    RUN("add/mad64", 256, 0, 2, 0, 1024);
    RUN("alignbit/mad64", 16, 1, 2, 0, 1024);
    RUN("add only", 8, 0, 0, 3, 1024);
    RUN("alignbit only", 8, 1, 1, 3, 1024);
    RUN("add/alignbit", 1, 0, 1, 3, 1024);
    RUN("add/alignbit", 16, 0, 1, 3, 1024);
    RUN("add/alignbit", 256, 0, 1, 3, 1024);
    RUN("add/alignbit", 1024, 0, 1, 3, 1024);
    RUN("add/alignbit", 16, 0, 1, 0, 512);
    RUN("add/alignbit", 256, 0, 1, 0, 512);
    RUN("add/alignbit", 1024, 0, 1, 0, 512);
    return 0;
}
