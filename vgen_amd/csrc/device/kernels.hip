// kernels.hip — the HIP kernels of libvgen_hip.so (gfx950 / CDNA4 only).
//
// seq_scan_kernel replaces the reference's single-pass search kernel `main`
// (src/shaders/search.wgsl:2-31): for every key k0 + i of a dispatch it produces the address payload
// (hash160 or Keccak address) — and, unlike the reference, applies the pattern prefilter on the
// device so that only candidates leave the GPU.
//
// Design (MI355X-first, not a translation of the WGSL):
//   * One inversion per WORKGROUP, not per key.  The reference spends 505 of its 521 field
//     multiplications per key in a per-thread Fermat inverse (field.wgsl:195-210).  Here every key is
//     an affine addition  Q_j (+/-) R_u  of a per-dispatch uniform point Q_j (S of them, read through
//     the scalar cache) and a per-lane table point R_u = (u*S + S/2)*G; the +R and -R results share
//     one denominator, each lane chains its S denominators into one product (prefix products parked
//     in LDS), the 256 lane products are combined in an LDS product tree, ONE Fermat inversion is
//     done per workgroup, and the tree is walked back down.  Cost per key: ~4.5 field
//     multiplications + 1/(512*S) of an inversion.
//   * Field arithmetic in 9x29-bit limbs on v_mad_u64_u32 with no carry flags (core/fe.h).
//   * SHA-256 / RIPEMD-160 / Keccak as straight-line register code (core/hash.h); no MFMA anywhere:
//     this is integer work bound by VALU issue, not a contraction.
//   * Output: dump mode writes 20 B per key (parity / reference-equivalent mode); filter mode
//     writes only candidate records through a wave-aggregated atomic slot counter.
#include <hip/hip_runtime.h>

#include "../core/ec.h"
#include "../core/filter_eval.h"
#include "../core/hash.h"
#include "device_types.h"
#include "launch.h"

namespace vg {

constexpr int WG = 256;

// ---- payload per format -----------------------------------------------------------------------------

template <int FMT>
__device__ __forceinline__ void payload_from_point(const u32 xw[8], const fe &y_canon, u32 out[5]) {
    if (FMT == VGF_P2PKH || FMT == VGF_P2WPKH) {
        u32 sha[8];
        sha256_pub33(2u | (y_canon.n[0] & 1u), xw, sha);
        ripemd160_of_sha(sha, out);
    } else if (FMT == VGF_P2SH_P2WPKH) {
        u32 sha[8], h[5];
        sha256_pub33(2u | (y_canon.n[0] & 1u), xw, sha);
        ripemd160_of_sha(sha, h);
        sha256_script22(h, sha);
        ripemd160_of_sha(sha, out);
    } else if (FMT == VGF_P2PKH_UNCOMPRESSED) {
        u32 yw[8], sha[8];
        fe_to_words(y_canon, yw);
        sha256_pub65(xw, yw, sha);
        ripemd160_of_sha(sha, out);
    } else {  // VGF_ETHEREUM
        u32 yw[8];
        fe_to_words(y_canon, yw);
        keccak256_pub64_addr(xw, yw, out);
    }
}

// ---- LDS helpers ---------------------------------------------------------------------------------------
// All LDS arrays are limb-major ([limb][lane]) so that a wave's access is 64 consecutive dwords.

__device__ __forceinline__ void lds_store_fe(u32 *base, int stride, int lane, const fe &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) base[i * stride + lane] = a.n[i];
}

__device__ __forceinline__ void lds_load_fe(const u32 *base, int stride, int lane, fe &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) a.n[i] = base[i * stride + lane];
}

// Inverts the WG per-lane values v (magnitude 1, non-zero) with ONE field inversion: an LDS product
// tree over the 256 lanes.  tree: 9 * 512 dwords, node k at column k (heap order: root = 1,
// leaves = 256..511).  Every thread of the workgroup must call this.
__device__ __forceinline__ void wg_batch_inverse(fe &v, u32 *tree) {
    const int tid = threadIdx.x;
    constexpr int ST = 2 * WG;
    lds_store_fe(tree, ST, WG + tid, v);
    __syncthreads();
    // up-sweep: node[k] = node[2k] * node[2k+1]
    for (int width = WG / 2; width >= 1; width >>= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe a, b, p;
            lds_load_fe(tree, ST, 2 * k, a);
            lds_load_fe(tree, ST, 2 * k + 1, b);
            fe_mul(p, a, b);
            lds_store_fe(tree, ST, k, p);
        }
        __syncthreads();
    }
    // root inverse, computed redundantly by the 64 lanes of wave 0 (same cost as one lane) so that
    // no lane-divergent branch wraps the 270-multiplication chain
    if (tid < 64) {
        fe r, ri;
        lds_load_fe(tree, ST, 1, r);
        fe_inv(ri, r);
        if (tid == 0) lds_store_fe(tree, ST, 1, ri);
    }
    __syncthreads();
    // down-sweep: inv[2k] = inv[k] * node[2k+1], inv[2k+1] = inv[k] * node[2k]
    for (int width = 1; width <= WG / 2; width <<= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe ik, a, b, ia, ib;
            lds_load_fe(tree, ST, k, ik);
            lds_load_fe(tree, ST, 2 * k, a);
            lds_load_fe(tree, ST, 2 * k + 1, b);
            fe_mul(ia, ik, b);
            fe_mul(ib, ik, a);
            lds_store_fe(tree, ST, 2 * k, ia);
            lds_store_fe(tree, ST, 2 * k + 1, ib);
        }
        __syncthreads();
    }
    lds_load_fe(tree, ST, WG + tid, v);
}

// ---- sequential-range scan ----------------------------------------------------------------------------

template <int FMT>
__global__ void __launch_bounds__(WG) seq_scan_kernel(const SeqArgs args) {
    extern __shared__ u32 lds[];
    const int tid = threadIdx.x;
    const u32 S = args.s;
    const u32 lanes = args.lanes;
    // LDS carve: prefix products [S][9][WG], then the inversion tree [9][2*WG]
    u32 *pre = lds;
    u32 *tree = lds + (size_t)S * 9 * WG;

    u32 u = blockIdx.x * WG + tid;
    const bool active = u < lanes;
    if (!active) u = lanes - 1;   // keep every lane in the workgroup-wide inversion; results discarded

    fe rx, ry, nrx, nry;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        rx.n[i] = args.rtab[(size_t)i * lanes + u];
        ry.n[i] = args.rtab[(size_t)(9 + i) * lanes + u];
    }
    fe_neg(nrx, rx, 1);   // magnitude 2
    fe_neg(nry, ry, 1);

    const DevSeqQ *__restrict__ q = args.q;

    // forward pass: acc = prod_j (R.x - Q_j.x); prefix products parked in LDS
    fe acc;
#pragma unroll 1
    for (u32 j = 0; j < S; j++) {
        fe dx;
#pragma unroll
        for (int i = 0; i < 9; i++) dx.n[i] = rx.n[i] + q[j].nqx[i];   // magnitude 2
        if (j == 0) {
            acc = dx;
            fe_normalize_weak(acc);
        } else {
            fe_mul(acc, acc, dx);
        }
        lds_store_fe(pre + (size_t)j * 9 * WG, WG, tid, acc);
    }

    // one inversion for the whole workgroup
    fe inv = acc;
    wg_batch_inverse(inv, tree);

    const u32 half = args.n >> 1;
    const bool dump = args.dump != nullptr;

    // backward pass
#pragma unroll 1
    for (int j = (int)S - 1; j >= 0; j--) {
        fe dx, idx;
#pragma unroll
        for (int i = 0; i < 9; i++) dx.n[i] = rx.n[i] + q[j].nqx[i];
        if (j > 0) {
            fe pj;
            lds_load_fe(pre + (size_t)(j - 1) * 9 * WG, WG, tid, pj);
            fe_mul(idx, inv, pj);
            fe_mul(inv, inv, dx);
        } else {
            idx = inv;
        }
#pragma unroll 1
        for (int sgn = 0; sgn < 2; sgn++) {
            // +R: dy = R.y - Q.y ; -R: dy = -R.y - Q.y
            fe dy, lam, x3, t, y3;
#pragma unroll
            for (int i = 0; i < 9; i++) dy.n[i] = (sgn ? nry.n[i] : ry.n[i]) + q[j].nqy[i];   // magnitude <= 3
            fe_mul(lam, dy, idx);
            fe_sqr(x3, lam);
#pragma unroll
            for (int i = 0; i < 9; i++) x3.n[i] += nrx.n[i] + q[j].nqx[i];                    // magnitude 4
            fe_normalize(x3);
            fe_neg(t, x3, 1);
#pragma unroll
            for (int i = 0; i < 9; i++) t.n[i] += q[j].qx[i];                                  // magnitude 3
            fe_mul(y3, lam, t);
#pragma unroll
            for (int i = 0; i < 9; i++) y3.n[i] += q[j].nqy[i];                                // magnitude 2
            fe_normalize(y3);

            u32 xw[8], pl[5];
            fe_to_words(x3, xw);
            payload_from_point<FMT>(xw, y3, pl);

            const u32 index = sgn ? (half - (u + 1) * S + (u32)j) : (half + u * S + (u32)j);
            if (active) {
                if (dump) {
                    u32 *o = args.dump + (size_t)index * 5;
#pragma unroll
                    for (int i = 0; i < 5; i++) o[i] = pl[i];
                } else if (filter_eval(args.filter, pl)) {
                    const u32 slot = atomicAdd(&args.mhdr->count, 1u);
                    if (slot < args.mhdr->cap) {
                        DevMatch *m = args.mrec + slot;
                        m->index = index;
                        m->reserved = 0;
#pragma unroll
                        for (int i = 0; i < 5; i++) m->payload[i] = pl[i];
                        m->payload[5] = m->payload[6] = m->payload[7] = 0;
                    }
                }
            }
        }
    }
}

// ---- launch wrappers (called from runtime.cpp) ----------------------------------------------------------

template <int FMT>
static hipError_t launch_seq_fmt(const SeqArgs &a, hipStream_t stream) {
    const u32 blocks = (a.lanes + WG - 1) / WG;
    const size_t lds_bytes = ((size_t)a.s * 9 * WG + 9 * 2 * WG) * sizeof(u32);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&seq_scan_kernel<FMT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(seq_scan_kernel<FMT>, dim3(blocks), dim3(WG), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_seq_scan(int fmt, const SeqArgs &a, hipStream_t stream) {
    switch (fmt) {
    case VGF_P2PKH:
    case VGF_P2WPKH:
        return launch_seq_fmt<VGF_P2PKH>(a, stream);
    case VGF_P2SH_P2WPKH:
        return launch_seq_fmt<VGF_P2SH_P2WPKH>(a, stream);
    case VGF_P2PKH_UNCOMPRESSED:
        return launch_seq_fmt<VGF_P2PKH_UNCOMPRESSED>(a, stream);
    case VGF_ETHEREUM:
        return launch_seq_fmt<VGF_ETHEREUM>(a, stream);
    default:
        return hipErrorInvalidValue;
    }
}

}  // namespace vg
