// core_dev.hip — TEST-ONLY device build of the single-source core (vgen_amd/csrc/core/*.h): the very functions
// the product kernels inline, run one input per lane on the GPU and returned to the test-suite, so that what
// exists only in the hipcc build is tested as compiled: the wave-uniform slow path behind VG_ANY_LANE
// (fe_canonicalize_product), the s_mov "opaque" multipliers of the column-form multiplication, v_bitop3 /
// v_alignbit forms of the hash rounds.  Not part of libvgen_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../vgen_amd/csrc/core/fe.h"
#include "../../vgen_amd/csrc/core/hash.h"

using namespace vg;

enum { OP_MUL, OP_SQR, OP_MUL_ADD, OP_SQR_ADD, OP_CANON_PRODUCT, OP_CANON, OP_PARITY_WEAK, OP_INV, OP_NORMALIZE,
       OP_NORMALIZE_WEAK, OP_NEG, OP_MUL_THEN_CANON_PRODUCT, OP_TO_WORDS, OP_INV_FERMAT };

__global__ void __launch_bounds__(64) fe_kernel(int op, int n, const u32 *a, const u32 *b, const u32 *c, u32 *r) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int k = i < n ? i : n - 1;   // every lane of the last wave computes (VG_ANY_LANE looks at all lanes)
    fe x, y, z, w;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        x.n[j] = a[(size_t)k * 9 + j];
        y.n[j] = b ? b[(size_t)k * 9 + j] : 0u;
        z.n[j] = c ? c[(size_t)k * 9 + j] : 0u;
    }
    fe_set_zero(w);
    switch (op) {
    case OP_MUL: fe_mul(w, x, y); break;
    case OP_SQR: fe_sqr(w, x); break;
    case OP_MUL_ADD: fe_mul_add(w, x, y, z); break;
    case OP_SQR_ADD: fe_sqr_add(w, x, z); break;
    case OP_CANON_PRODUCT: w = x; fe_canonicalize_product(w); break;
    case OP_CANON: w = x; fe_canonicalize(w); break;
    case OP_PARITY_WEAK: w.n[0] = fe_parity_weak(x); break;
    case OP_INV: fe_inv(w, x); break;
    case OP_INV_FERMAT: fe_inv_fermat(w, x); break;
    case OP_NORMALIZE: w = x; fe_normalize(w); break;
    case OP_NORMALIZE_WEAK: w = x; fe_normalize_weak(w); break;
    case OP_NEG: fe_neg(w, x, y.n[0]); break;
    case OP_MUL_THEN_CANON_PRODUCT: fe_mul(w, x, y); fe_canonicalize_product(w); break;
    case OP_TO_WORDS: {
        u32 ww[8];
        fe_to_words(x, ww);
        fe_from_words(w, ww);
        break;
    }
    }
    if (i < n) {
#pragma unroll
        for (int j = 0; j < 9; j++) r[(size_t)i * 9 + j] = w.n[j];
    }
}

enum { H_PUB33, H_PUB65, H_SCRIPT22, H_KECCAK };

__device__ inline void be32_to_words(const unsigned char *be, u32 w[8]) {
    for (int i = 0; i < 8; i++)
        w[i] = ((u32)be[28 - 4 * i] << 24) | ((u32)be[29 - 4 * i] << 16) | ((u32)be[30 - 4 * i] << 8) | be[31 - 4 * i];
}

// x_be / y_be: 32 big-endian bytes per item (H_SCRIPT22: x_be holds the 20-byte hash160 in its first 20 bytes);
// prefix: H_PUB33's 02/03 per item.  out: 20 bytes per item.
__global__ void __launch_bounds__(64) hash_kernel(int op, int n, const unsigned char *x_be, const unsigned char *y_be,
                                                  const unsigned char *prefix, unsigned char *out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 xw[8], yw[8], sha[8], h[5];
    be32_to_words(x_be + 32 * (size_t)i, xw);
    if (y_be) be32_to_words(y_be + 32 * (size_t)i, yw);
    if (op == H_PUB33) {
        sha256_pub33(prefix[i], xw, sha);
        ripemd160_of_sha(sha, h);
    } else if (op == H_PUB65) {
        sha256_pub65(xw, yw, sha);
        ripemd160_of_sha(sha, h);
    } else if (op == H_SCRIPT22) {
        const unsigned char *p = x_be + 32 * (size_t)i;
        u32 hin[5];
        for (int j = 0; j < 5; j++) hin[j] = (u32)p[4 * j] | ((u32)p[4 * j + 1] << 8) | ((u32)p[4 * j + 2] << 16) | ((u32)p[4 * j + 3] << 24);
        sha256_script22(hin, sha);
        ripemd160_of_sha(sha, h);
    } else {
        keccak256_pub64_addr(xw, yw, h);
    }
    for (int j = 0; j < 5; j++)
        for (int b = 0; b < 4; b++) out[20 * (size_t)i + 4 * j + b] = (unsigned char)(h[j] >> (8 * b));
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -(int)e_ - 1000; } while (0)

extern "C" {

int coredev_device_count() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int coredev_fe(int op, int n, const u32 *a, const u32 *b, const u32 *c, u32 *r) {
    if (n <= 0) return 0;
    const size_t bytes = (size_t)n * 9 * sizeof(u32);
    u32 *da = nullptr, *db = nullptr, *dc = nullptr, *dr = nullptr;
    CK(hipMalloc((void **)&da, bytes));
    CK(hipMalloc((void **)&dr, bytes));
    CK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    if (b) {
        CK(hipMalloc((void **)&db, bytes));
        CK(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
    }
    if (c) {
        CK(hipMalloc((void **)&dc, bytes));
        CK(hipMemcpy(dc, c, bytes, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(fe_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, op, n, da, db, dc, dr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r, dr, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dr);
    return 0;
}

int coredev_hash(int op, int n, const unsigned char *x_be, const unsigned char *y_be, const unsigned char *prefix, unsigned char *out) {
    if (n <= 0) return 0;
    unsigned char *dx = nullptr, *dy = nullptr, *dp = nullptr, *d_out = nullptr;
    CK(hipMalloc((void **)&dx, (size_t)n * 32));
    CK(hipMemcpy(dx, x_be, (size_t)n * 32, hipMemcpyHostToDevice));
    if (y_be) {
        CK(hipMalloc((void **)&dy, (size_t)n * 32));
        CK(hipMemcpy(dy, y_be, (size_t)n * 32, hipMemcpyHostToDevice));
    }
    if (prefix) {
        CK(hipMalloc((void **)&dp, (size_t)n));
        CK(hipMemcpy(dp, prefix, (size_t)n, hipMemcpyHostToDevice));
    }
    CK(hipMalloc((void **)&d_out, (size_t)n * 20));
    hipLaunchKernelGGL(hash_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, op, n, dx, dy, dp, d_out);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, (size_t)n * 20, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dp); (void)hipFree(d_out);
    return 0;
}

}
