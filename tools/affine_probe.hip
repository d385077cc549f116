// affine_probe.hip — MEASURES the batched-affine form of the per-key scalar multiplication (VERDICT r03 "next" #6).
//
// Today's arbitrary-scalar path (keys_fwd_kernel, what the reference's CPU loop does per candidate with libsecp256k1's
// ec_pubkey_create, src/scanner.rs:151-155) accumulates the 11 window points of k*G in Jacobian coordinates in REGISTERS:
// 10 mixed additions of 8 M + 3 S = 11 field multiplications each, 21 300 VALU instructions per key, issue-bound at 662 us
// per 2^20 keys (profiles/pmc_keys.json).  An AFFINE addition costs 1 M + 1 S + 1 M once 1/(x2 - x1) is known, and the
// inverses of a whole batch cost ~3 M each when they share one inversion (Montgomery's trick: prefix products per lane, a
// product tree per workgroup, one root per workgroup inverted by seq_inv_kernel's divsteps): ~6.5 M per addition, a 1.6x
// ceiling — IF the accumulators of all keys can wait somewhere while the shared inversion happens.  EXPERIMENTS.md (round 3) rejected
// this on a traffic estimate (the accumulators and prefix products park in HBM: ~3 KB per key); this probe builds it and
// measures it, with the fusion that gives the idea its best shot:
//
//   aff_first    acc = T_0[d_0] (a gather, no arithmetic); denominators of window 1, per-lane prefix products, tree, root
//   seq_inv      the roots of all workgroups, one per lane                                           } x 10 windows
//   aff_step(w)  tree down, per key: peel 1/dx, lambda, x3, y3 (the addition of window w) — and, fused, }
//                the denominators / prefix products / tree / root of window w + 1
//   (last step)  writes the affine public keys
//
// B keys per lane (B = 8: the per-lane chain of seq_fwd_kernel), key j of lane u = index j * lanes + u (coalesced);
// parked per key: acc x, y (18 limbs) and one prefix product (9 limbs) = 108 B written and read per window, plus two
// 64-byte table sectors gathered.  Checked against core/ec.h's Jacobian multiplication (same 24-bit table) on every key.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/affine_probe.hip -o tools/affine_probe \
//              -L vgen_amd -lvgen_hip -Wl,-rpath,'$ORIGIN/../vgen_amd' -Wl,-rpath,/opt/rocm/lib
// Run:   tools/affine_probe [log2 keys = 20] [streams = 1,4]      (profiles/r04_affine_probe.txt; counters: tools/affine_probe_pmc.sh)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../vgen_amd/csrc/core/ec.h"
#include "../vgen_amd/csrc/core/rnd.h"
#include "../vgen_amd/csrc/device/device_types.h"
#include "../vgen_amd/csrc/device/launch.h"
#include "../vgen_amd/csrc/host/host_ec.h"

using namespace vg;

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int WG = 256;
constexpr int B = 8;            // keys per lane
constexpr u32 WB = 24;          // window width: the product's default table
constexpr u32 NWIN = 11;
constexpr unsigned long long NE = (1ull << WB) - 1ull;

struct AffArgs {
    const u32 *tab;       // wide table: [NWIN][2^24 - 1][16 words]
    const u32 *keys;      // [n][8] little-endian words (already range-checked on the host side of the probe)
    u32 *acc;             // [18][n] limb-major: x limbs 0..8, y limbs 9..17
    u32 *pre;             // [9][n] limb-major: prefix product up to and including this key (in the lane's chain order)
    u32 *tree;            // [groups][9][WG]
    u32 *root;            // [9][groups]
    u32 *out;             // [n][16]: x, y as eight little-endian words each (last step)
    u32 lanes, groups, n;
    u32 w;                // the window whose addition this step finishes (aff_step)
};

__device__ __forceinline__ void lds_store_fe(u32 *base, int stride, int col, const fe &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) base[i * stride + col] = a.n[i];
}
__device__ __forceinline__ void lds_load_fe(const u32 *base, int stride, int col, fe &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) a.n[i] = base[i * stride + col];
}

// table point of window w for scalar k (digit 0: entry 1 stands in, `zero` says so)
__device__ __forceinline__ void gather_entry(const AffArgs &a, const u32 k[8], u32 w, bool xy, fe &tx, fe &ty, bool &zero) {
    const u32 d = ec_wide_digit(k, w, WB);
    zero = d == 0;
    const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(a.tab + ((unsigned long long)w * NE + ((d ? d : 1u) - 1u)) * 16ull);
    u32 raw[16];
#pragma unroll
    for (int q = 0; q < (xy ? 4 : 2); q++) {
        const ec_u4 t4 = e4[q];
#pragma unroll
        for (int i = 0; i < 4; i++) raw[4 * q + i] = t4.v[i];
    }
    fe_from_words(tx, raw);
    if (xy) fe_from_words(ty, raw + 8);
}

__device__ __forceinline__ void load_key(const AffArgs &a, u32 idx, u32 k[8]) {
    const ec_u4 *p = reinterpret_cast<const ec_u4 *>(a.keys + (size_t)idx * 8);
    const ec_u4 lo = p[0], hi = p[1];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        k[i] = lo.v[i];
        k[4 + i] = hi.v[i];
    }
}

// product tree over the workgroup's lane products, tree and root written out (as seq_fwd_kernel)
__device__ __forceinline__ void tree_up(u32 *tree, const fe &lane_prod, const AffArgs &a) {
    const int tid = threadIdx.x;
    fe sib, pair;
#pragma unroll
    for (int i = 0; i < 9; i++) sib.n[i] = (u32)__shfl_xor((int)lane_prod.n[i], 1);
    fe_mul(pair, lane_prod, sib);
    if ((tid & 1) == 0) lds_store_fe(tree, WG, WG / 2 + (tid >> 1), pair);
    __syncthreads();
#pragma unroll 1
    for (int width = WG / 4; width >= 1; width >>= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe x, y, p;
            lds_load_fe(tree, WG, 2 * k, x);
            lds_load_fe(tree, WG, 2 * k + 1, y);
            fe_mul(p, x, y);
            lds_store_fe(tree, WG, k, p);
        }
        __syncthreads();
    }
    u32 *tg = a.tree + (size_t)blockIdx.x * 9 * WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tg[i * WG + tid] = tree[i * WG + tid];
    if (tid < 9) a.root[(size_t)tid * a.groups + blockIdx.x] = tree[tid * WG + 1];
}

// ... and back: 1 / (this lane's product), from the inverted root
__device__ __forceinline__ void tree_down(u32 *tree, const AffArgs &a, const fe &sib_prod, fe &inv) {
    const int tid = threadIdx.x;
    const u32 *tg = a.tree + (size_t)blockIdx.x * 9 * WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tree[i * WG + tid] = tg[i * WG + tid];
    __syncthreads();
    if (tid < 9) tree[tid * WG + 1] = a.root[(size_t)tid * a.groups + blockIdx.x];
    __syncthreads();
#pragma unroll 1
    for (int width = 1; width <= WG / 4; width <<= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe ik, x, y, ix, iy;
            lds_load_fe(tree, WG, k, ik);
            lds_load_fe(tree, WG, 2 * k, x);
            lds_load_fe(tree, WG, 2 * k + 1, y);
            fe_mul(ix, ik, y);
            fe_mul(iy, ik, x);
            lds_store_fe(tree, WG, 2 * k, ix);
            lds_store_fe(tree, WG, 2 * k + 1, iy);
        }
        __syncthreads();
    }
    fe ip;
    lds_load_fe(tree, WG, WG / 2 + (tid >> 1), ip);
    fe_mul(inv, ip, sib_prod);
    __syncthreads();   // the tree's LDS is reused by tree_up of the next window
}

// acc = T_0[d_0]; denominators of window 1
__global__ void __launch_bounds__(WG) aff_first_kernel(const AffArgs a) {
    __shared__ u32 tree[9 * WG];
    const u32 u = blockIdx.x * WG + threadIdx.x;
    fe run;
#pragma unroll 1
    for (int j = 0; j < B; j++) {
        const u32 idx = (u32)j * a.lanes + u;
        u32 k[8];
        load_key(a, idx, k);
        fe ax, ay, tx, ty;
        bool z0, z1;
        gather_entry(a, k, 0, true, ax, ay, z0);
        if (z0) fe_set_zero(ay);                 // y = 0: "still at infinity" (no point of this curve has y = 0)
        gather_entry(a, k, 1, false, tx, ty, z1);
        fe dx;
        fe_sub_n(dx, tx, ax);
        if (z0 || z1) fe_set_one(dx);            // nothing to add (or nothing to add to): keep the chain invertible
        if (j == 0) run = dx;
        else fe_mul(run, run, dx);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            a.acc[(size_t)i * a.n + idx] = ax.n[i];
            a.acc[(size_t)(9 + i) * a.n + idx] = ay.n[i];
            a.pre[(size_t)i * a.n + idx] = run.n[i];
        }
    }
    tree_up(tree, run, a);
}

// the addition of window a.w for every key, fused with the denominators of window a.w + 1 (LAST: writes the keys instead).
// The lane's chain runs j = 0..B-1 in even windows' forward passes and is peeled in the opposite order; the NEXT window's
// chain is built in that peeling order, so directions alternate: DIR = +1 peels B-1..0, DIR = -1 peels 0..B-1.
template <bool LAST>
__global__ void __launch_bounds__(WG) aff_step_kernel(const AffArgs a, int dir) {
    __shared__ u32 tree[9 * WG];
    const u32 u = blockIdx.x * WG + threadIdx.x;
    // the neighbour lane's product = its last prefix product in chain order
    const int last_j = dir > 0 ? B - 1 : 0;
    fe sib, inv;
    {
        const u32 sidx = (u32)last_j * a.lanes + (u ^ 1u);
#pragma unroll
        for (int i = 0; i < 9; i++) sib.n[i] = a.pre[(size_t)i * a.n + sidx];
    }
    tree_down(tree, a, sib, inv);     // inv = 1 / (this lane's product of dx)
    fe run;                           // next window's chain
    bool first = true;
#pragma unroll 1
    for (int s = 0; s < B; s++) {
        const int j = dir > 0 ? B - 1 - s : s;            // peel from the chain's end
        const int jprev = dir > 0 ? j - 1 : j + 1;        // the key before it in chain order
        const u32 idx = (u32)j * a.lanes + u;
        u32 k[8];
        load_key(a, idx, k);
        fe ax, ay, tx, ty;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            ax.n[i] = a.acc[(size_t)i * a.n + idx];
            ay.n[i] = a.acc[(size_t)(9 + i) * a.n + idx];
        }
        bool zt;
        gather_entry(a, k, a.w, true, tx, ty, zt);
        u32 ynz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) ynz |= ay.n[i];
        const bool inf = ynz == 0;
        const bool skip = zt || inf;
        fe dx, idx_inv;
        fe_sub_n(dx, tx, ax);
        if (skip) fe_set_one(dx);
        if (s < B - 1) {
            fe pp;
            const u32 pidx = (u32)jprev * a.lanes + u;
#pragma unroll
            for (int i = 0; i < 9; i++) pp.n[i] = a.pre[(size_t)i * a.n + pidx];
            fe_mul(idx_inv, inv, pp);
            fe_mul(inv, inv, dx);
        } else {
            idx_inv = inv;
        }
        // lambda = (ty - ay) / (tx - ax); x3 = lambda^2 - ax - tx; y3 = lambda (ax - x3) - ay
        fe dy, lam, x3, y3, t;
        fe_sub_n(dy, ty, ay);
        fe_mul(lam, dy, idx_inv);
        fe_sqr(x3, lam);
        fe_sub_n(x3, x3, ax);
        fe_sub_n(x3, x3, tx);
        fe_sub_n(t, ax, x3);
        fe_mul(y3, lam, t);
        fe_sub_n(y3, y3, ay);
        // selects: a zero digit keeps acc; "still at infinity" takes the table point
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const u32 nx = inf ? tx.n[i] : x3.n[i], ny = inf ? ty.n[i] : y3.n[i];
            ax.n[i] = zt ? ax.n[i] : nx;
            ay.n[i] = zt ? ay.n[i] : ny;
        }
        if (LAST) {
            fe_canonicalize(ax);
            fe_canonicalize(ay);
            u32 xw[8], yw[8];
            fe_to_words(ax, xw);
            fe_to_words(ay, yw);
            ec_u4 *o = reinterpret_cast<ec_u4 *>(a.out + (size_t)idx * 16);
            o[0] = ec_u4{{xw[0], xw[1], xw[2], xw[3]}};
            o[1] = ec_u4{{xw[4], xw[5], xw[6], xw[7]}};
            o[2] = ec_u4{{yw[0], yw[1], yw[2], yw[3]}};
            o[3] = ec_u4{{yw[4], yw[5], yw[6], yw[7]}};
        } else {
            // next window's denominator, chained in THIS pass's order
            fe nx_, ny_;
            bool zn;
            gather_entry(a, k, a.w + 1, false, nx_, ny_, zn);
            u32 y2 = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) y2 |= ay.n[i];
            fe ndx;
            fe_sub_n(ndx, nx_, ax);
            if (zn || y2 == 0) fe_set_one(ndx);
            if (first) run = ndx;
            else fe_mul(run, run, ndx);
            first = false;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                a.acc[(size_t)i * a.n + idx] = ax.n[i];
                a.acc[(size_t)(9 + i) * a.n + idx] = ay.n[i];
                a.pre[(size_t)i * a.n + idx] = run.n[i];
            }
        }
    }
    if (!LAST) tree_up(tree, run, a);
}

__global__ void __launch_bounds__(64) inv_kernel(u32 *root, u32 groups) {
    __builtin_amdgcn_s_setprio(3);
    const u32 g = blockIdx.x * 64 + threadIdx.x;
    const u32 gg = g < groups ? g : groups - 1;
    fe r, ri;
#pragma unroll
    for (int i = 0; i < 9; i++) r.n[i] = root[(size_t)i * groups + gg];
    fe_inv(ri, r);
    if (g < groups) {
#pragma unroll
        for (int i = 0; i < 9; i++) root[(size_t)i * groups + g] = ri.n[i];
    }
}

// reference: the product's Jacobian multiplication over the same table + the lane's own inversion (slow; correctness only)
__global__ void __launch_bounds__(WG) ref_kernel(const u32 *tab, const u32 *keys, u32 *out, u32 n) {
    const u32 idx = blockIdx.x * WG + threadIdx.x;
    if (idx >= n) return;
    u32 k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = keys[(size_t)idx * 8 + i];
    gej acc;
    ec_mul_gen_wide<24>(acc, k, tab);
    fe zi, zi2, zi3, x, y;
    fe_inv(zi, acc.z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(x, acc.x, zi2);
    fe_mul(y, acc.y, zi3);
    fe_canonicalize(x);
    fe_canonicalize(y);
    u32 xw[8], yw[8];
    fe_to_words(x, xw);
    fe_to_words(y, yw);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[(size_t)idx * 16 + i] = xw[i];
        out[(size_t)idx * 16 + 8 + i] = yw[i];
    }
}

// timing twin of keys_fwd_kernel: the Jacobian accumulation alone, results parked as the product parks them (27 limbs)
__global__ void __launch_bounds__(WG) jac_kernel(const u32 *tab, const u32 *keys, u32 *xyz, u32 n) {
    const u32 idx = blockIdx.x * WG + threadIdx.x;
    if (idx >= n) return;
    u32 k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = keys[(size_t)idx * 8 + i];
    gej acc;
    ec_mul_gen_wide<24>(acc, k, tab);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        xyz[(size_t)i * n + idx] = acc.x.n[i];
        xyz[(size_t)(9 + i) * n + idx] = acc.y.n[i];
        xyz[(size_t)(18 + i) * n + idx] = acc.z.n[i];
    }
}

// keys: SHA-256 counter stream (core/rnd.h), as little-endian words, clamped into [1, n) by clearing the top bit
__global__ void __launch_bounds__(WG) keys_kernel(u32 *keys, u32 n, u32 salt) {
    const u32 idx = blockIdx.x * WG + threadIdx.x;
    if (idx >= n) return;
    u32 k[8];
    rnd_scalar(rnd_seed_from_u64(0xA11CEull + salt), 7, idx, 0, k);
    k[7] &= 0x7FFFFFFFu;          // < 2^255 < n
    k[0] |= 1u;                   // never zero
#pragma unroll
    for (int i = 0; i < 8; i++) keys[(size_t)idx * 8 + i] = k[i];
}

struct Batch {
    AffArgs a;
    hipStream_t st;
};

static void enqueue_affine(Batch &b) {
    AffArgs a = b.a;
    hipLaunchKernelGGL(aff_first_kernel, dim3(a.groups), dim3(WG), 0, b.st, a);
    int dir = +1;                    // aff_first chained j = 0..B-1: the first peel runs B-1..0
    for (u32 w = 1; w < NWIN; w++) {
        hipLaunchKernelGGL(inv_kernel, dim3((a.groups + 63) / 64), dim3(64), 0, b.st, a.root, a.groups);
        a.w = w;
        if (w + 1 < NWIN) hipLaunchKernelGGL((aff_step_kernel<false>), dim3(a.groups), dim3(WG), 0, b.st, a, dir);
        else hipLaunchKernelGGL((aff_step_kernel<true>), dim3(a.groups), dim3(WG), 0, b.st, a, dir);
        dir = -dir;
    }
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const u32 n = 1u << lg;
    const u32 lanes = n / B, groups = lanes / WG;
    if (lg < 12 || lg > 22 || lanes % WG) {
        fprintf(stderr, "log2 keys must be 12..22\n");
        return 1;
    }
    // tables: 8-bit on the host, 24-bit on the device (the product's own builders)
    std::vector<u32> t8;
    host_gen_table8_limbs(t8);
    u32 *d_t8, *d_tab, *d_small;
    CHECK(hipMalloc(&d_t8, t8.size() * 4));
    CHECK(hipMemcpy(d_t8, t8.data(), t8.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_tab, ec_wide_words(WB) * 4ull));
    CHECK(hipMalloc(&d_small, ec_wide_words(WB / 2) * 4ull));
    CHECK(launch_gen_table_wide(d_t8, d_tab, d_small, WB, 0));
    CHECK(hipDeviceSynchronize());

    const int max_streams = 4;
    std::vector<Batch> bs(max_streams);
    for (int s = 0; s < max_streams; s++) {
        AffArgs &a = bs[s].a;
        memset(&a, 0, sizeof a);
        u32 *keys, *acc, *pre, *tree, *root, *out;
        CHECK(hipMalloc(&keys, (size_t)n * 32));
        CHECK(hipMalloc(&acc, (size_t)n * 72));
        CHECK(hipMalloc(&pre, (size_t)n * 36));
        CHECK(hipMalloc(&tree, (size_t)groups * 9 * WG * 4));
        CHECK(hipMalloc(&root, (size_t)groups * 9 * 4));
        CHECK(hipMalloc(&out, (size_t)n * 64));
        a.tab = d_tab; a.keys = keys; a.acc = acc; a.pre = pre; a.tree = tree; a.root = root; a.out = out;
        a.lanes = lanes; a.groups = groups; a.n = n;
        CHECK(hipStreamCreateWithFlags(&bs[s].st, hipStreamNonBlocking));
        hipLaunchKernelGGL(keys_kernel, dim3(n / WG), dim3(WG), 0, 0, keys, n, (u32)s);   // DISTINCT scalars per batch
    }
    CHECK(hipDeviceSynchronize());

    // ---- correctness: every key of batch 0 against the Jacobian path ----
    u32 *d_ref;
    CHECK(hipMalloc(&d_ref, (size_t)n * 64));
    hipLaunchKernelGGL(ref_kernel, dim3(n / WG), dim3(WG), 0, 0, d_tab, bs[0].a.keys, d_ref, n);
    enqueue_affine(bs[0]);
    CHECK(hipDeviceSynchronize());
    {
        std::vector<u32> r((size_t)n * 16), g((size_t)n * 16);
        CHECK(hipMemcpy(r.data(), d_ref, r.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(g.data(), bs[0].a.out, g.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, firstbad = 0;
        for (size_t i = 0; i < n; i++)
            if (memcmp(&r[i * 16], &g[i * 16], 64) != 0 && bad++ == 0) firstbad = i;
        printf("{\"check\":\"affine vs Jacobian (same table), all %u keys\",\"mismatches\":%zu,\"first\":%zu}\n", n, bad, firstbad);
        if (bad) return 2;
    }
    CHECK(hipFree(d_ref));

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // ---- the Jacobian accumulation (keys_fwd_kernel's work), alone on the chip ----
    {
        u32 *xyz;
        CHECK(hipMalloc(&xyz, (size_t)n * 108));
        hipLaunchKernelGGL(jac_kernel, dim3(n / WG), dim3(WG), 0, 0, d_tab, bs[0].a.keys, xyz, n);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(jac_kernel, dim3(n / WG), dim3(WG), 0, 0, d_tab, bs[rep % max_streams].a.keys, xyz, n);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("{\"path\":\"jacobian accumulation in registers (keys_fwd_kernel's work), one launch\",\"keys\":%u,\"ms\":%.4f,\"Mkeys_per_s\":%.1f}\n", n, best,
               n / (best * 1e-3) / 1e6);
        CHECK(hipFree(xyz));
    }
    // ---- batched affine: one batch alone (latency of the 21-kernel chain), then 2 and 4 batches in flight on streams of their own ----
    for (int ns : {1, 2, 4}) {
        const int rounds = 6;
        for (int s = 0; s < ns; s++) enqueue_affine(bs[s]);   // warm
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            CHECK(hipStreamWaitEvent(bs[0].st, e0, 0));
            for (int r = 0; r < rounds; r++)
                for (int s = 0; s < ns; s++) enqueue_affine(bs[s]);
            for (int s = 0; s < ns; s++) CHECK(hipStreamSynchronize(bs[s].st));
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double per_batch = best / (rounds * ns);
        printf("{\"path\":\"batched affine, window by window (fused step kernels), %d batch(es) in flight\",\"keys\":%u,\"ms_per_batch\":%.4f,\"Mkeys_per_s\":%.1f,"
               "\"parked_bytes_per_key_per_window\":%d}\n", ns, n, per_batch, n / (per_batch * 1e-3) / 1e6, 2 * (72 + 36));
    }
    return 0;
}
