// ubench_valu.hip — integer-VALU issue-rate microbenchmark for gfx950.
//
// The scan path is bound by integer VALU throughput; its roofline peak was an
// estimate there ("verify by microbenchmark on the box").  This program measures, per instruction
// class, the sustained lane-ops/s of the whole chip at 1/2/4/8 waves per SIMD, so that the peak in
// bench.py's roofline object and the limb-width choice of the field multiplier rest on
// measurements.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int CHAINS = 8;    // independent dependency chains per lane
constexpr int UNROLL = 32;   // ops per chain per loop trip

enum Op { ADD, ADDC, ADD3, MUL_LO, MUL_HI, MAD64, MAD24, MULHI24, ROTXOR, BFI, XOR3, FMA32, FMA64, LSHLADD, PERM, XOR, LSHLADD64, ADDCO, LSHR64, LSHL64, AND, LSHR32, MOV, BFE, ANDOR, LSHLOR, CNDMASK, MULU24, MADCHAIN, MIX_ADD_ALIGN, MIX_ADD2_ALIGN, MIX_BITOP_ALIGN, MIX_ADD_MAD, MIX_ALT1, MIX_RUN32 };

template <int OP>
__global__ void __launch_bounds__(256) k_bench(uint32_t *out, uint32_t seed, int iters, unsigned long long *clk) {
    // shader-clock (s_memtime) and constant 100 MHz (s_memrealtime) stamps around the loop: their ratio is the
    // clock the SIMDs really ran at under this instruction mix (power management lowers it below the nominal 2.4 GHz)
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t x[CHAINS], y = seed | 1, z = seed * 2654435761u + 12345u;
    uint64_t acc[CHAINS];
    float f[CHAINS];
    double d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        x[c] = threadIdx.x * 747796405u + c * 2891336453u + seed;
        acc[c] = ((uint64_t)x[c] << 32) | (x[c] ^ 0x9E3779B9u);
        f[c] = 1.0f + (x[c] & 1023) * 1e-6f;
        d[c] = 1.0 + (x[c] & 1023) * 1e-9;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
#define A3(ins) asm volatile(ins " %0, %1, %2" : "=v"(x[c]) : "v"(x[c]), "v"(y))
#define A4(ins) asm volatile(ins " %0, %1, %2, %3" : "=v"(x[c]) : "v"(x[c]), "v"(y), "v"(z))
                if (OP == ADD) A3("v_add_u32");
                if (OP == ADDC) {  // 64-bit add as the compiler emits it for carry chains
                    uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
                    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                                 : "+v"(lo), "+v"(hi) : "v"(y), "v"(z) : "vcc");
                    acc[c] = ((uint64_t)hi << 32) | lo;
                }
                if (OP == ADD3) A4("v_add3_u32");
                if (OP == MUL_LO) A3("v_mul_lo_u32");
                if (OP == MUL_HI) A3("v_mul_hi_u32");
                if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x[c]), "v"(y) : "vcc");
                if (OP == MAD24) A4("v_mad_u32_u24");
                if (OP == MULHI24) A3("v_mul_hi_u32_u24");
                if (OP == ROTXOR) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[c]));
                if (OP == BFI) A4("v_bfi_b32");
                if (OP == XOR3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[c]) : "v"(y), "v"(z));
                if (OP == FMA32) f[c] = __builtin_fmaf(f[c], 1.0000001f, 1e-7f);
                if (OP == FMA64) d[c] = __builtin_fma(d[c], 1.0000001, 1e-9);
                if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(y));
                if (OP == PERM) A4("v_perm_b32");
                if (OP == XOR) A3("v_xor_b32");
                if (OP == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
                if (OP == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[c]) : "v"(y) : "vcc");
                if (OP == LSHR64) asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
                if (OP == LSHL64) asm volatile("v_lshlrev_b64 %0, 8, %1" : "=v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
                if (OP == AND) A3("v_and_b32");
                if (OP == LSHR32) asm volatile("v_lshrrev_b32 %0, 29, %1" : "=v"(x[c]) : "v"(x[(c + 1) % CHAINS]));
                if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[c]) : "v"(x[(c + 1) % CHAINS]));
                if (OP == BFE) asm volatile("v_bfe_u32 %0, %1, 3, 29" : "=v"(x[c]) : "v"(x[(c + 1) % CHAINS]));
                if (OP == ANDOR) A4("v_and_or_b32");
                if (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(y));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(y) : "vcc");
                if (OP == MULU24) A3("v_mul_u32_u24");
                if (OP == MIX_ADD_ALIGN) { if (u & 1) A3("v_add_u32"); else asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[c])); }
                if (OP == MIX_ADD2_ALIGN) { if (u % 3) A3("v_add_u32"); else asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[c])); }
                if (OP == MIX_BITOP_ALIGN) { if (u & 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[c]) : "v"(y), "v"(z)); else asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[c])); }
                if (OP == MIX_ADD_MAD) { if (u & 1) A3("v_add_u32"); else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x[c]), "v"(y) : "vcc"); }
                if (OP == MIX_ALT1) { if ((u + c) & 1) A3("v_add_u32"); else asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[c])); }
                if (OP == MIX_RUN32) { if (u & 4) A3("v_add_u32"); else asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[c])); }
                if (OP == MADCHAIN) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(x[c]), "v"(y) : "vcc");
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r ^= x[c] ^ (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32) ^ __float_as_uint(f[c]) ^ (uint32_t)__double_as_longlong(d[c]);
    if (r == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = r;  // practically never
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int OP>
static double run(const char *name, int waves_per_simd, int iters, uint32_t *dout, double instr_per_op) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    static unsigned long long *dclk = nullptr;
    if (!dclk) CHECK(hipMalloc(&dclk, 16));
    hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u, iters / 8, dclk);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u, iters, dclk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256.0 * (double)iters * UNROLL * CHAINS;
    double tops = ops / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double wave_instr_per_simd = (double)waves_per_simd * iters * UNROLL * CHAINS * instr_per_op;
    double cyc = (ms * 1e-3 * 2.4e9) / wave_instr_per_simd;
    unsigned long long clk[2];
    CHECK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
    const double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0.0;   // s_memrealtime ticks at 100 MHz
    const double true_cyc = (double)clk[0] / ((double)waves_per_simd * iters * UNROLL * CHAINS * instr_per_op);
    printf("{\"op\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.3f,\"Tlaneops\":%.2f,\"cyc_per_waveinstr_at_2.4GHz\":%.2f,"
           "\"shader_mhz\":%.0f,\"shader_cycles_per_waveinstr\":%.2f}\n",
           name, waves_per_simd, ms, tops, cyc, mhz, true_cyc);
    fflush(stdout);
    return tops;
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    uint32_t *dout;
    CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(uint32_t) * 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\":\"%s\",\"cus\":%d,\"clock_khz\":%d,\"lds_per_block\":%zu,\"regs_per_block\":%d}\n", prop.name,
           prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock, prop.regsPerBlock);
    int ws[] = {1, 4, 8};
    for (int w : ws) {
        run<ADD>("v_add_u32", w, iters, dout, 1);
        run<ADDC>("add64(v_add_co+v_addc)", w, iters, dout, 2);
        run<ADD3>("v_add3_u32", w, iters, dout, 1);
        run<MUL_LO>("v_mul_lo_u32", w, iters, dout, 1);
        run<MUL_HI>("v_mul_hi_u32", w, iters, dout, 1);
        run<MAD64>("v_mad_u64_u32", w, iters, dout, 1);
        run<MAD24>("v_mad_u32_u24", w, iters, dout, 1);
        run<MULHI24>("v_mul_hi_u32_u24", w, iters, dout, 1);
        run<ROTXOR>("v_alignbit_b32", w, iters, dout, 1);
        run<BFI>("v_bfi_b32", w, iters, dout, 1);
        run<XOR3>("v_bitop3_b32", w, iters, dout, 1);
        run<XOR>("v_xor_b32", w, iters, dout, 1);
        run<LSHLADD64>("v_lshl_add_u64", w, iters, dout, 1);
        run<ADDCO>("v_add_co_u32", w, iters, dout, 1);
        run<LSHLADD>("v_lshl_add_u32", w, iters, dout, 1);
        run<PERM>("v_perm_b32", w, iters, dout, 1);
        run<FMA32>("v_pk_fma_f32(2 fma/instr)", w, iters, dout, 0.5);
        run<FMA64>("v_fma_f64", w, iters, dout, 1);
        run<LSHR64>("v_lshrrev_b64", w, iters, dout, 1);
        run<LSHL64>("v_lshlrev_b64", w, iters, dout, 1);
        run<AND>("v_and_b32", w, iters, dout, 1);
        run<LSHR32>("v_lshrrev_b32", w, iters, dout, 1);
        run<MOV>("v_mov_b32", w, iters, dout, 1);
        run<BFE>("v_bfe_u32", w, iters, dout, 1);
        run<ANDOR>("v_and_or_b32", w, iters, dout, 1);
        run<LSHLOR>("v_lshl_or_b32", w, iters, dout, 1);
        run<CNDMASK>("v_cndmask_b32", w, iters, dout, 1);
        run<MULU24>("v_mul_u32_u24", w, iters, dout, 1);
        run<MADCHAIN>("v_mad_u64_u32(single dependent chain)", w, iters, dout, 1);
        run<MIX_ADD_ALIGN>("mix 1:1 v_add_u32 / v_alignbit_b32 (same chain alternates per op)", w, iters, dout, 1);
        run<MIX_ADD2_ALIGN>("mix 2:1 v_add_u32 / v_alignbit_b32", w, iters, dout, 1);
        run<MIX_BITOP_ALIGN>("mix 1:1 v_bitop3_b32 / v_alignbit_b32", w, iters, dout, 1);
        run<MIX_ADD_MAD>("mix 1:1 v_add_u32 / v_mad_u64_u32", w, iters, dout, 1);
        run<MIX_ALT1>("mix 1:1 v_add_u32 / v_alignbit_b32, alternating every instruction", w, iters, dout, 1);
        run<MIX_RUN32>("mix 1:1 v_add_u32 / v_alignbit_b32, runs of 32 instructions", w, iters, dout, 1);
    }
    CHECK(hipFree(dout));
    return 0;
}
