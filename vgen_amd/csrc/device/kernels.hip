// kernels.hip — the HIP kernels of libvgen_hip.so (gfx950 / CDNA4 only).
//
// Together the three seq_* kernels replace the reference's single-pass search kernel `main`
// (src/shaders/search.wgsl:2-31): for every key k0 + i of a dispatch they produce the address payload
// (hash160 or Keccak address) — and, unlike the reference, apply the pattern prefilter on the device
// so that only candidates leave the GPU.
//
// Design (MI355X-first, not a translation of the WGSL):
//   * One field inversion per 64 WORKGROUPS' worth of keys instead of one per key.  The reference
//     spends 505 of its 521 field multiplications per key in a per-thread Fermat inverse
//     (field.wgsl:195-210).  Here every key is an affine addition  Q_j (+/-) R_u  of a per-dispatch
//     uniform point Q_j (S of them, passed as kernel arguments and read through the scalar cache) and
//     a per-lane table point R_u = (u*S + S/2)*G.  The +R and -R results share one denominator; each
//     lane chains its S denominators (prefix products parked in L2-resident global scratch); the 256
//     lane products of a workgroup are combined by an LDS product tree whose ROOT is written out
//     (seq_fwd_kernel).  seq_inv_kernel then inverts the roots of all workgroups with one root per
//     LANE — an inversion (divsteps, core/fe.h) is a serial dependency no matter what, so it only pays
//     when all 64 lanes of the wave carry different roots.  seq_bwd_kernel walks the trees back down
//     and runs the per-key tail.  Cost per key: ~4.5 field multiplications; the inversion is noise.
//     (A fused single kernel with the inversion inside was measured first: the lone inverting wave
//     stalled its workgroup for 0.2 ms of a 0.33 ms dispatch — profiles/r01_*.)
//   * Field arithmetic in 9x29-bit limbs on v_mad_u64_u32 with no carry flags (core/fe.h).
//   * SHA-256 / RIPEMD-160 / Keccak as straight-line register code (core/hash.h); no MFMA anywhere:
//     this is integer work bound by VALU issue, not a contraction.
//   * Output: dump mode writes 20 B per key (parity / reference-equivalent mode); filter mode writes
//     only candidate records through a wave-aggregated atomic slot counter.
//   * The same staging (per-lane work -> product tree -> one root per lane inverted -> finish) carries the two
//     paths that need a scalar multiplication per key: arbitrary scalars (keys_fwd / keys_bwd) and the taproot
//     tweak (seq_bwd<P2TR> parks Q = P + t*G, p2tr_finish_kernel completes it; behind arbitrary scalars the same work
//     runs one key per lane: p2tr_tweak_kernel / p2tr_out_kernel).  Fused forms with the inversion inside one kernel were
//     measured first and lost a third to a half to the lone inverting wave.
//   * One launch is ~1 wave per SIMD at the default batch, so the device is filled by frames in flight (twelve by
//     default, one stream each: runtime.cpp); seq_bwd is capped at 128 VGPRs so that four launches share a SIMD.
//   * The waves of the short, latency-bound first half of a dispatch (seq_fwd_kernel, seq_inv_kernel) raise their issue
//     priority (s_setprio 3) over the seq_bwd waves of other frames they share a SIMD with — same instruction count, but
//     the chain of the NEXT dispatch is not slowed to a quarter of its speed by the arbiter's round robin
//     (profiles/r02_queue_sweep.txt: 11.4 instead of 9.5 Gkeys/s at four frames).
#include <hip/hip_runtime.h>

#ifndef VG_PARK
#define VG_PARK 0
#endif
#ifndef VG_KECCAK_BLOCK
#define VG_KECCAK_BLOCK 1   // Ethereum: Keccak-f as a generated block in runs by issue class (payload_from_point)
#endif
#ifndef VG_PARKI
#define VG_PARKI 1
#endif
#define VG_HASH_BLOCKS 1   // core/dfa_eval.h: base58_checksum runs as a scheduled block too (hash_blocks.inc below)
#include "../core/dfa_eval.h"
#include "../core/ec.h"
#include "../core/filter_eval.h"
#include "../core/hash.h"
#include "../core/rnd.h"
#include "../core/taproot.h"
#include "device_types.h"
#include "launch.h"

namespace vg {

constexpr int WG = SEQ_WG;   // 256 lanes per workgroup

#ifndef VG_SEQ_WAVES_P2TR
#define VG_SEQ_WAVES_P2TR 2
#endif
#ifndef VG_SEQ_WAVES_ETH
#define VG_SEQ_WAVES_ETH 4
#endif

// ---- LDS / lane helpers ---------------------------------------------------------------------------------
// All arrays are limb-major ([limb][lane]) so that a wave's access is 64 consecutive dwords.

__device__ __forceinline__ void lds_store_fe(u32 *base, int stride, int col, const fe &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) base[i * stride + col] = a.n[i];
}

__device__ __forceinline__ void lds_load_fe(const u32 *base, int stride, int col, fe &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) a.n[i] = base[i * stride + col];
}

// A field element parked in LDS across a register-hungry stretch.  Volatile, so that the compiler does not keep a copy
// in registers and forward it to the reload — and through an explicit LDS address-space pointer: a volatile access
// through a generic pointer compiles to flat_store / flat_load with system scope and a 64-bit address per limb.
typedef __attribute__((address_space(3))) volatile u32 lds_vu32;

__device__ __forceinline__ void lds_park_fe(u32 *base, int stride, int col, const fe &a) {
    lds_vu32 *b = (lds_vu32 *)base;
#pragma unroll
    for (int i = 0; i < 9; i++) b[i * stride + col] = a.n[i];
}

__device__ __forceinline__ void lds_unpark_fe(u32 *base, int stride, int col, fe &a) {
    lds_vu32 *b = (lds_vu32 *)base;
#pragma unroll
    for (int i = 0; i < 9; i++) a.n[i] = b[i * stride + col];
}

// beta, the cube root of unity mod p behind the secp256k1 endomorphism lambda * (x, y) = (beta * x, y), in 29-bit limbs
__device__ __forceinline__ void fe_set_beta(fe &b) {
    b.n[0] = 0x119501EEu; b.n[1] = 0x09CB6143u; b.n[2] = 0x1D626570u; b.n[3] = 0x0092EA25u; b.n[4] = 0x034E99CFu;
    b.n[5] = 0x03CF561Au; b.n[6] = 0x1C41B991u; b.n[7] = 0x056CAF80u; b.n[8] = 0x007AE96Au;
}

__device__ __forceinline__ void shfl_xor_fe(fe &r, const fe &a, int mask) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.n[i] = (u32)__shfl_xor((int)a.n[i], mask);
}

// ---- match compaction: one atomic per WAVE ------------------------------------------------------------
// The candidates of a wave are compacted with the wave's own ballot (north_star: "wavefront ballot/reduce for match
// compaction"): the lanes that hold a candidate are counted (s_bcnt1 of the ballot mask), the first of them reserves that
// many consecutive slots of the frame's ring with ONE global atomic, and every candidate takes the reserved base plus its
// rank among the wave's candidates (v_mbcnt of the mask).  Written out here rather than left to LLVM's atomic optimizer,
// which happened to produce this form from a per-lane atomicAdd(…, 1): tests/test_isa_contract.py pins the result in the ISA.
// Must be reached by all lanes of the wave that are still active together (it is: `hit` is computed by every such lane).
// -> the slot relative to this dispatch's base for lanes with hit (monotonic counter, no per-dispatch reset), unspecified for the others.
__device__ __forceinline__ u32 match_slot(DevMatchHeader *hdr, bool hit, u32 match_base) {
    const unsigned long long m = __ballot(hit);
    if (m == 0) return 0xFFFFFFFFu;            // wave-uniform: the common case costs the ballot and one scalar branch
    const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));   // candidates in lower lanes
    const int leader = __ffsll((long long)m) - 1;
    u32 first = 0;
    if (hit && rank == 0) first = atomicAdd(&hdr->count, (u32)__popcll(m));
    first = (u32)__builtin_amdgcn_readlane((int)first, leader);
    return first + rank - match_base;
}

// ---- the address hashes as scheduled instruction blocks ----------------------------------------------------
// hash160_pub33_block / hash160_script22_block / hash160_pub65_block / base58_check_block: one asm statement each, generated by
// device/hashgen.py from the same round functions as core/hash.h (the Makefile writes hash_blocks.inc).  The order of the
// instructions — runs by issue class — and the wave-priority changes between the runs are part of the kernel's design, not hipcc's.
#include "hash_blocks.inc"

// ---- payload per format -----------------------------------------------------------------------------

// What a FULL kernel tells the on-device matcher about the address format at compile time (core/dfa_eval.h: the encoders that
// cannot be needed are not compiled in: Base58Check +3 % with its checksum as a scheduled block, A/B profiles/r04_hash_blocks_ab.txt).
// Not for Ethereum: the kernel that carries only the hex walk measured 1.8 % SLOWER than the one that still carries all three
// (7.61 against 7.75 Gkeys/s, twice; another register allocation of the Keccak rounds), so that format keeps the run-time choice.
template <int FMT>
struct MatchFmt {
    static constexpr int value = (FMT == VGF_P2TR || FMT == VGF_ETHEREUM) ? -1 : FMT;
};

template <int FMT>
struct PayloadWords {
    static constexpr int value = (FMT == VGF_P2TR) ? 8 : 5;
};

// x, y: canonical affine public key.  out: the payload words in memory order (PayloadWords<FMT>), for the five formats
// whose payload is a hash of the key itself (the taproot output key needs a scalar multiplication and a shared inversion
// of its own: p2tr_tweak_kernel / p2tr_out_kernel below).
// LONE: hipcc's own schedule of core/hash.h instead of the block — for launches that have the chip to themselves (one wave per
// SIMD), where the compiler's interleaving of rounds wins and every priority change costs four cycles (111 against ~130 us at 2^20 keys).
template <int FMT, bool LONE = false>
__device__ __forceinline__ bool payload_from_point(const fe &x, const fe &y_canon, u32 *out) {
    static_assert(FMT != VGF_P2TR, "taproot payloads come from p2tr_tweak_kernel / p2tr_out_kernel");
    static_assert(!LONE || FMT == VGF_P2PKH, "the lone variant exists for the headline format only");
    u32 xw[8];
    fe_to_words(x, xw);
    if (LONE) {
        u32 sha[8];
        sha256_pub33(2u | (y_canon.n[0] & 1u), xw, sha);
        ripemd160_of_sha(sha, out);
    } else if (FMT == VGF_P2PKH || FMT == VGF_P2WPKH) {
        hash160_pub33_block(2u | (y_canon.n[0] & 1u), xw, out);
    } else if (FMT == VGF_P2SH_P2WPKH) {
        u32 h[5];
        hash160_pub33_block(2u | (y_canon.n[0] & 1u), xw, h);
        hash160_script22_block(h, out);
    } else if (FMT == VGF_P2PKH_UNCOMPRESSED) {
        u32 yw[8];
        fe_to_words(y_canon, yw);
        hash160_pub65_block(xw, yw, out);
    } else {  // VGF_ETHEREUM
        u32 yw[8];
        fe_to_words(y_canon, yw);
#if VG_KECCAK_BLOCK   // Keccak-f as one generated block of 4 195 instructions in runs by issue class (device/hashgen.py): +28 % on every Ethereum
        // configuration (profiles/r05_keccak_ab.txt).  In dependency order without the priority changes (round 4) it only matched hipcc's rolled rounds.
        keccak_addr_block(xw, yw, out);
#else
        keccak256_pub64_addr(xw, yw, out);
#endif
    }
    return true;
}

// Scratch layout (global, one region per frame), for a dispatch of `lanes` lanes in lanes/WG groups:
//   pre  : [S][9][lanes]        prefix products p_0 .. p_{S-1} of every lane (p_{S-1} = lane product)
//   tree : [groups][9][WG]      product-tree nodes of each workgroup, heap order: columns 1..WG/2-1
//                               internal, WG/2 .. WG-1 the pair products; column 0 unused
//   root : [9][groups]          tree roots (seq_fwd) -> their inverses in place (seq_inv)

// ---- stage 1: denominators, per-lane products, workgroup product tree ---------------------------------

__global__ void __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(1, 8)))
seq_fwd_kernel(const SeqArgs args) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ u32 tree[9 * WG];
    const int tid = threadIdx.x;
    const u32 S = args.s;
    const u32 lanes = args.lanes;
    const u32 u = blockIdx.x * WG + tid;   // lanes is a multiple of WG

    fe rx;
#pragma unroll
    for (int i = 0; i < 9; i++) rx.n[i] = args.rtab[(size_t)i * lanes + u];

    u32 *pre = args.pre + u;
    fe acc;
#pragma unroll 1
    for (u32 j = 0; j < S; j++) {
        fe dx;
#pragma unroll
        for (int i = 0; i < 9; i++) dx.n[i] = rx.n[i] + args.q[j].nqx[i];   // R.x - Q_j.x, magnitude 2
        if (j == 0) {
            acc = dx;
            fe_normalize_weak(acc);
        } else {
            fe_mul(acc, acc, dx);
        }
#pragma unroll
        for (int i = 0; i < 9; i++) pre[(size_t)(j * 9 + i) * lanes] = acc.n[i];
    }

    // product tree; the leaf level is exchanged by a lane shuffle
    fe sib, pair;
    shfl_xor_fe(sib, acc, 1);
    fe_mul(pair, acc, sib);
    if ((tid & 1) == 0) lds_store_fe(tree, WG, WG / 2 + (tid >> 1), pair);
    __syncthreads();
    for (int width = WG / 4; width >= 1; width >>= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe a, b, p;
            lds_load_fe(tree, WG, 2 * k, a);
            lds_load_fe(tree, WG, 2 * k + 1, b);
            fe_mul(p, a, b);
            lds_store_fe(tree, WG, k, p);
        }
        __syncthreads();
    }
    // write the tree out (coalesced) and the root into the root vector
    u32 *tg = args.tree + (size_t)blockIdx.x * 9 * WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tg[i * WG + tid] = tree[i * WG + tid];
    if (tid < 9) args.root[(size_t)tid * args.groups + blockIdx.x] = tree[tid * WG + 1];
}

// ---- stage 2: invert every workgroup's root, one root per lane -----------------------------------------

__global__ void __launch_bounds__(64) seq_inv_kernel(u32 *root, u32 groups) {
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

// ---- stage 3: walk the tree down, finish the additions, hash, filter ---------------------------------------

// FULL: the filter is the pattern's whole DFA (DEVF_DFA): its tables are staged into dynamic LDS and every
// key's address is encoded and matched on the device (core/dfa_eval.h).  A separate instantiation so
// that the extra registers of the Base58Check path do not touch the prefilter kernels' occupancy.
// Register budget: 4 waves per SIMD (128 VGPRs) for the 20-byte formats — four launches of different frames
// then share a SIMD; Keccak (50 state registers) runs better at 3 waves without spills (6.16 vs 5.85 Gkeys/s),
// the taproot path (a full scalar multiplication per key) keeps its ~240 registers (217 vs 197 Mkeys/s).
#ifndef VG_SEQ_WAVES_P2PKH
#define VG_SEQ_WAVES_P2PKH 4
#endif
#ifndef VG_SEQ_WAVES_P2SH
#define VG_SEQ_WAVES_P2SH 4
#endif
#ifndef VG_SEQ_WAVES_UNCOMP
#define VG_SEQ_WAVES_UNCOMP 4
#endif
#ifndef VG_SEQ_WAVES_FULL20
#define VG_SEQ_WAVES_FULL20 4
#endif
#ifndef VG_SEQ_WAVES_EC
#define VG_SEQ_WAVES_EC 4      // seq_bwd_kernel<.., SPLIT>: the point arithmetic alone
#endif
#ifndef VG_SEQ_WAVES_HASH
#define VG_SEQ_WAVES_HASH 8    // seq_hash_kernel: the generated hash block needs 28 registers + the key's nine words
#endif
// Wave priority by issue class (round 5; tools/ubench_phase*.hip, tools/issue_model.py): a SIMD fills each 4-cycle issue slot with the next
// instruction of its highest-priority ready wave and, behind it, one FULL-RATE instruction of another wave.  Half-rate instructions and the
// multiply-adds of the point arithmetic only ever take the first place, so the scan kernels run at priority 1 and the generated hash blocks drop
// to 0 for their runs of full-rate instructions (device/hashgen.py), which then ride in the second places: 13.2 -> 15.5 Gkeys/s.
#ifndef VG_BASE_PRIO
#define VG_BASE_PRIO 1
#endif
#ifndef VG_EC_PRIO
#define VG_EC_PRIO 0
#endif
#ifndef VG_EC_SPRIO
#define VG_EC_SPRIO 0
#endif
#ifndef VG_SEQ_WAVES_HASH_FULL
#define VG_SEQ_WAVES_HASH_FULL 6
#endif
template <int FMT, bool FULL, bool ENDO = false>
struct SeqWaves {
    // (Ethereum: four waves since the Keccak block and the running inverse in LDS; the six-image on-device matcher would spill there: three)
    static constexpr int value = FMT == VGF_P2TR ? VG_SEQ_WAVES_P2TR : FMT == VGF_ETHEREUM ? (FULL && ENDO ? 3 : VG_SEQ_WAVES_ETH)
                                 : FULL ? VG_SEQ_WAVES_FULL20 : FMT == VGF_P2SH_P2WPKH ? VG_SEQ_WAVES_P2SH
                                 : FMT == VGF_P2PKH_UNCOMPRESSED ? VG_SEQ_WAVES_UNCOMP : VG_SEQ_WAVES_P2PKH;
};

// (The blocks that write a payload to the dump or the match ring, and the six-image loops of seq_bwd_kernel and keys_bwd_kernel,
//  are written out in place on purpose: round 3 folded them into two shared __forceinline__ helpers — same instruction counts,
//  2 787 VALU per key in the headline loop — and the Ethereum kernels lost 1-2.5 % to a different scalar-register allocation
//  (one form re-read args.dump inside the per-key loop and waited for the scalar load; A/B on one box, tools/ab_fmt.sh).)
//
// ENDO (vanity searches; every format but P2TR, with a prefilter, the on-device DFA or in dump mode): every point is tested under its six
// endomorphism / negation images — (x, +-y), (beta x, +-y), (beta^2 x, +-y), the public keys of k, lambda k, lambda^2 k and
// their negations — so six keys are hashed for one point's arithmetic plus two multiplications by beta.  Image `variant`
// = s * 3 + e (e = power of beta, s = negated) of key index i is reported / dumped at variant * n + i.
// LONE (P2PKH / P2WPKH with a prefilter only): the variant for dispatches issued while at most one other frame of the context is in flight
// (runtime.cpp: rt_dispatch), see payload_from_point.
// SPLIT (the compressed-key formats without ENDO): the kernel stops at the affine point — it parks the eight words of x and the
// key's prefix byte (0x02 | parity of y) per key in args.xs and leaves hashes, filter and output to seq_hash_kernel, which runs one
// key per lane at eight waves per SIMD (see there).
template <int FMT, bool FULL, bool ENDO = false, bool LONE = false, bool SPLIT = false>
__global__ void __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(SPLIT ? VG_SEQ_WAVES_EC : SeqWaves<FMT, FULL, ENDO>::value, SPLIT ? VG_SEQ_WAVES_EC : SeqWaves<FMT, FULL, ENDO>::value)))
seq_bwd_kernel(const SeqArgs args) {
#if VG_BASE_PRIO
    // (the one-frame twin too: its launches share SIMDs with the steady-state kernel's whenever a burst of dispatches starts, and at priority 0 they
    //  would only ever be served when no other wave has anything to issue - the in-order host loop then waits on them: second completion of a burst
    //  at 831 us instead of ~540)
    if (!SPLIT) __builtin_amdgcn_s_setprio(VG_BASE_PRIO);
#endif
#if VG_EC_PRIO
    if (SPLIT) __builtin_amdgcn_s_setprio(VG_EC_PRIO);   // the point arithmetic has the memory bubbles: it goes first when it can issue, the hash waves fill in
#endif
    static_assert(!SPLIT || (!FULL && !ENDO && !LONE && FMT == VGF_P2PKH), "the split form parks compressed keys: x and the parity of y");
    __shared__ u32 tree[9 * WG];
    __shared__ u32 ypark[ENDO && (FMT == VGF_P2PKH_UNCOMPRESSED || FMT == VGF_ETHEREUM) ? 9 * WG : 1];   // ENDO: the point's y
    extern __shared__ u32 dyn_lds[];   // FULL: the DFA blob
    constexpr int NW = PayloadWords<FMT>::value;
    // PARK: the lane's table point R_u and the running inverse live in LDS between their uses instead of in 27 registers (LDS reads
    // issue beside the VALU, not through it), so that the prefilter kernels of the compressed-key formats fit VG_SEQ_WAVES_P2PKH > 4
    // PARK2 (VG_PARK=2): nothing of the lane's state stays in registers across a key — R_u is re-read from the offset table (L2-resident, coalesced)
    // where it is needed, the running inverse and the step's 1/dx wait in LDS (18 KB per workgroup: eight workgroups share a CU).
    constexpr int PARKM = (!FULL && !ENDO && !LONE && (FMT == VGF_P2PKH || FMT == VGF_P2WPKH)) ? VG_PARK : 0;
    constexpr bool PARK = PARKM == 1, PARK2 = PARKM == 2;
    // PARKI: the instantiations that would otherwise spill a few registers at their 128-register cap (the uncompressed-key format: a second
    // SHA-256 block's sixteen message words; the six-image on-device matcher) keep the running inverse in 9 KB of LDS of its own instead
    constexpr bool PARKI = VG_PARKI && !PARK && !PARK2 && !LONE && (FMT == VGF_P2PKH_UNCOMPRESSED || FMT == VGF_ETHEREUM || (FMT == VGF_P2PKH && FULL && ENDO));
    __shared__ u32 rpark[PARK ? 17 * WG : (PARK2 || PARKI) ? 9 * WG : 1];   // (PARK: R.y's top limb stays in a register: 26 KB of LDS per workgroup lets six share a CU)
    const int tid = threadIdx.x;
    const GenTables gtab{args.gtab, args.gtab16, args.gtab_bits};   // P2TR: fixed-window generator tables, read from global memory (L2 / Infinity Cache / HBM)
    u32 *dfa_lds = dyn_lds;
    if (FULL) {
        for (u32 i = tid; i < args.dfa_bytes / 4; i += WG) dfa_lds[i] = args.dfa_blob[i];
        // (visibility: the barriers of the tree phase below come before the first read of the table)
    }
    const u32 S = args.s;
    const u32 lanes = args.lanes;
    const u32 u = blockIdx.x * WG + tid;
    // shader-clock sample: the first wave of the launch reads the shader-clock counter (s_memtime) and the constant
    // 100 MHz counter (s_memrealtime) when it starts and when it ends, and adds both spans to the frame's match
    // header — the clock the CUs really ran at while this very kernel executed, at the cost of four scalar reads
    const bool stamp = FMT != VGF_P2TR && !SPLIT && blockIdx.x == 0 && args.mhdr != nullptr;
    const unsigned long long stamp_c0 = stamp ? clock64() : 0ull, stamp_w0 = stamp ? wall_clock64() : 0ull;

    const u32 *tg = args.tree + (size_t)blockIdx.x * 9 * WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tree[i * WG + tid] = tg[i * WG + tid];
    __syncthreads();
    if (tid < 9) tree[tid * WG + 1] = args.root[(size_t)tid * args.groups + blockIdx.x];   // root^-1
    __syncthreads();
    // down-sweep: inv[2k] = inv[k] * node[2k+1], inv[2k+1] = inv[k] * node[2k]
    for (int width = 1; width <= WG / 4; width <<= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe ik, a, b, ia, ib;
            lds_load_fe(tree, WG, k, ik);
            lds_load_fe(tree, WG, 2 * k, a);
            lds_load_fe(tree, WG, 2 * k + 1, b);
            fe_mul(ia, ik, b);
            fe_mul(ib, ik, a);
            lds_store_fe(tree, WG, 2 * k, ia);
            lds_store_fe(tree, WG, 2 * k + 1, ib);
        }
        __syncthreads();
    }
    const u32 *pre = args.pre + u;
    fe inv;
    {
        // 1/acc = 1/(acc * sib) * sib, sib = the neighbouring lane's product p_{S-1}
        fe ip, sib;
        lds_load_fe(tree, WG, WG / 2 + (tid >> 1), ip);
#pragma unroll
        for (int i = 0; i < 9; i++) sib.n[i] = args.pre[(size_t)((S - 1) * 9 + i) * lanes + (u ^ 1u)];
        fe_mul(inv, ip, sib);
    }
    if (ENDO || PARK || PARK2) __syncthreads();   // every lane has read its pair's inverse: the tree's LDS now parks the x of the point in hand (PARK: the running inverse)

    fe rx, ry;
    if (!PARK2) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            rx.n[i] = args.rtab[(size_t)i * lanes + u];
            ry.n[i] = args.rtab[(size_t)(9 + i) * lanes + u];
        }
    }
    if (PARK2) lds_park_fe(tree, WG, tid, inv);
    if (PARKI) lds_park_fe(rpark, WG, tid, inv);
    if (PARK) {
        lds_park_fe(rpark, WG, tid, rx);
        {
            lds_vu32 *b = (lds_vu32 *)(rpark + 9 * WG);
#pragma unroll
            for (int i = 0; i < 8; i++) b[i * WG + tid] = ry.n[i];
        }
        lds_park_fe(tree, WG, tid, inv);
    }

    const u32 half = args.n >> 1;
    const bool dump = args.dump != nullptr;
    fe zrun;            // P2TR: running product of this lane's Z(Q)
    u32 step = 0;       // P2TR: key step 0 .. 2S-1 in loop order

#pragma unroll 1
    for (int j = (int)S - 1; j >= 0; j--) {
        const DevSeqQ &q = args.q[j];
        fe dx, idx;
        if (PARK) {
            lds_unpark_fe(rpark, WG, tid, rx);
            lds_unpark_fe(tree, WG, tid, inv);
        }
        if (PARKI) lds_unpark_fe(rpark, WG, tid, inv);
        if (PARK2) {
            const u32 *rt = args.rtab + u;
            asm volatile("" : "+v"(rt));   // (an address the compiler cannot recognise as the loop invariant it is: the loads stay here)
#pragma unroll
            for (int i = 0; i < 9; i++) rx.n[i] = rt[(size_t)i * lanes];
            lds_unpark_fe(tree, WG, tid, inv);
        }
#pragma unroll
        for (int i = 0; i < 9; i++) dx.n[i] = rx.n[i] + q.nqx[i];
        if (j > 0) {
            fe pj;
#pragma unroll
            for (int i = 0; i < 9; i++) pj.n[i] = pre[(size_t)((j - 1) * 9 + i) * lanes];
            fe_mul(idx, inv, pj);
            fe_mul(inv, inv, dx);
            if (PARK || PARK2) lds_park_fe(tree, WG, tid, inv);
            if (PARKI) lds_park_fe(rpark, WG, tid, inv);
        } else {
            idx = inv;
        }
        if (PARK2) lds_park_fe(rpark, WG, tid, idx);
        // -R.x and -R.y are recomputed where needed (9 subtractions each) instead of living in 18 registers
        // across the loop; the empty asm keeps the compiler from hoisting them back out as loop invariants.
        if (!PARK && !PARK2) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                asm volatile("" : "+v"(rx.n[i]));
                asm volatile("" : "+v"(ry.n[i]));
            }
        }
        fe nsum, nqy;   // -(R.x + Q.x) (magnitude 3) and -Q.y, shared by the +R and -R results
        if (!PARK2) fe_neg(nsum, rx, 1);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            if (!PARK2) nsum.n[i] += q.nqx[i];
            nqy.n[i] = q.nqy[i];
        }
#pragma unroll 1
        for (int sgn = 0; sgn < 2; sgn++) {
            // +R: dy = R.y - Q.y ; -R: dy = -R.y - Q.y
            fe dy, lam, x3, t, y3;
            if (PARK2) {
                const u32 *rt = args.rtab + u;
                asm volatile("" : "+v"(rt));
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    rx.n[i] = rt[(size_t)i * lanes];
                    ry.n[i] = rt[(size_t)(9 + i) * lanes];
                }
                fe_neg(nsum, rx, 1);
#pragma unroll
                for (int i = 0; i < 9; i++) nsum.n[i] += q.nqx[i];
                lds_unpark_fe(rpark, WG, tid, idx);
            }
            if (PARK) {
                lds_vu32 *b = (lds_vu32 *)(rpark + 9 * WG);
#pragma unroll
                for (int i = 0; i < 8; i++) ry.n[i] = b[i * WG + tid];
            }
            if (sgn) fe_neg(dy, ry, 1);
            else dy = ry;
#pragma unroll
            for (int i = 0; i < 9; i++) dy.n[i] += q.nqy[i];   // magnitude <= 3
            fe_mul(lam, dy, idx);
            fe_sqr_add(x3, lam, nsum);            // lam^2 - R.x - Q.x, weakly normalised
            // VG_EC_SPRIO (A/B): the carry chains between the multiplications are runs of full-rate instructions: at priority 0 they can ride
            // behind other waves' half-rate instructions like the hash blocks' full-rate runs do
            if (VG_EC_SPRIO && !LONE) __builtin_amdgcn_s_setprio(0);
            fe_canonicalize_product(x3);
            fe_neg(t, x3, 1);
#pragma unroll
            for (int i = 0; i < 9; i++) t.n[i] += q.qx[i];                                  // magnitude 3
            if (VG_EC_SPRIO && !LONE) __builtin_amdgcn_s_setprio(VG_BASE_PRIO);
            fe_mul_add(y3, lam, t, nqy);          // lam*(Q.x - x3) - Q.y
            if (VG_EC_SPRIO && !LONE) __builtin_amdgcn_s_setprio(0);
            if (FMT == VGF_P2PKH || FMT == VGF_P2WPKH || FMT == VGF_P2SH_P2WPKH)
                y3.n[0] = fe_parity_weak(y3);     // a compressed key takes only the parity of y (bit 0 is all that is read)
            else
                fe_canonicalize_product(y3);

            if (FMT == VGF_P2TR) {
                // Taproot, stage A: the tweaked point Q = lift_x(x) + t*G stays Jacobian; X, Z and the lane's
                // running product of Z's are parked for the second shared inversion (p2tr_finish_kernel).  [The same stage as kernels
                // of its own — p2tr_tweak_kernel at three waves per SIMD instead of this kernel's two — measured SLOWER here
                // (1.21 vs 1.26 Gkeys/s): the chip holds ~2.07 GHz under this multiplier-dense code whatever the occupancy, so only
                // the instruction count matters, and one key per lane pays the workgroup's product tree per key instead of per
                // 2S keys; that form serves the arbitrary-scalar path, which has one key per lane anyway.]
                gej qq;
                const bool okq = taproot_tweak_point(x3, y3, gtab, qq);
                const bool zero = taproot_z_is_zero(qq.z);   // t*G == -P: no address; keep the products invertible
                if (zero) fe_set_one(qq.z);
                if (step == 0) zrun = qq.z;
                else fe_mul(zrun, zrun, qq.z);
                u32 *o = args.tq + (size_t)step * 27 * lanes + u;
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    o[(size_t)i * lanes] = qq.x.n[i];
                    o[(size_t)(9 + i) * lanes] = qq.z.n[i];
                    o[(size_t)(18 + i) * lanes] = zrun.n[i];
                }
                args.tq_flag[(size_t)step * lanes + u] = (okq && !zero) ? 1u : 0u;
                step++;
                continue;
            }

            if (SPLIT) {
                // key step (2j + sgn) of lane u -> slot (2j + sgn) * lanes + u of the dispatch, word-major: a wave's stores are 64 consecutive dwords
                u32 xw[8];
                fe_to_words(x3, xw);
                u32 *o = args.xs + (size_t)((u32)j * 2u + (u32)sgn) * lanes + u;
#pragma unroll
                for (int i = 0; i < 8; i++) o[(size_t)i * args.n] = xw[i];
                o[(size_t)8 * args.n] = 2u | (y3.n[0] & 1u);
                continue;
            }
            const u32 index = sgn ? (half - (u + 1) * S + (u32)j) : (half + u * S + (u32)j);
            if (ENDO) {
                // compressed-key formats need only the parity of y (flipped for the negations); the others the canonical
                // y itself, parked beside x, and p - y for the negations
                constexpr bool NEEDS_Y = FMT == VGF_P2PKH_UNCOMPRESSED || FMT == VGF_ETHEREUM;
                const u32 ypar = y3.n[0] & 1u;
                lds_park_fe(tree, WG, tid, x3);
                if (NEEDS_Y) lds_park_fe(ypark, WG, tid, y3);
#pragma unroll 1
                for (u32 v = 0; v < 6; v++) {
                    const u32 e = v >> 1, sneg = v & 1u;   // (x,+) (x,-) (bx,+) (bx,-) (b^2 x,+) (b^2 x,-)
                    fe xe, ye;
                    lds_unpark_fe(tree, WG, tid, xe);
                    if (sneg == 0 && e > 0) {
                        fe beta;
                        fe_set_beta(beta);
                        fe_mul(xe, xe, beta);
                        fe_canonicalize_product(xe);
                        lds_park_fe(tree, WG, tid, xe);
                    }
                    if (NEEDS_Y) {
                        lds_unpark_fe(ypark, WG, tid, ye);
                        if (sneg) {
                            fe ny;
                            fe_neg(ny, ye, 1);
                            fe_normalize(ny);      // p - y, canonical (y != 0 on this curve)
                            ye = ny;
                        }
                    } else {
                        ye.n[0] = ypar ^ sneg;     // all a compressed key reads of y
                    }
                    u32 ple[NW];
                    (void)payload_from_point<FMT == VGF_P2TR ? VGF_P2PKH : FMT>(xe, ye, ple);
                    const u32 vindex = (sneg * 3u + e) * args.n + index;
                    if (dump) {
                        u32 *o = args.dump + (size_t)vindex * NW;
#pragma unroll
                        for (int i = 0; i < NW; i++) o[i] = ple[i];
                    } else {
                        const bool hit = FULL ? dfa_match_payload_n<NW, MatchFmt<FMT>::value>(dfa_lds, (int)args.fmt, ple) : filter_eval_n<NW>(args.filter, ple);
                        const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
                        if (hit && slot < args.match_cap) {
                            DevMatch *m = args.mrec + slot;
                            m->index = vindex;
                            m->reserved = 0;
#pragma unroll
                            for (int i = 0; i < 8; i++) m->payload[i] = i < NW ? ple[i] : 0u;
                        }
                    }
                }
                continue;
            }

            u32 pl[NW];
            const bool ok = payload_from_point<FMT == VGF_P2TR ? VGF_P2PKH : FMT, LONE>(x3, y3, pl);   // (never reached for P2TR: see above)

            if (dump) {
                u32 *o = args.dump + (size_t)index * NW;
#pragma unroll
                for (int i = 0; i < NW; i++) o[i] = ok ? pl[i] : 0u;
            } else {
                // monotonic counter: no per-dispatch reset; this dispatch's slots start at match_base
                const bool hit = ok && (FULL ? dfa_match_payload_n<NW, MatchFmt<FMT>::value>(dfa_lds, (int)args.fmt, pl) : filter_eval_n<NW>(args.filter, pl));
                const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
                if (hit && slot < args.match_cap) {
                    DevMatch *m = args.mrec + slot;
                    m->index = index;
                    m->reserved = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) m->payload[i] = i < NW ? pl[i] : 0u;
                }
            }
        }
    }
    if (stamp && tid == 0) {   // (launches of one frame are ordered on their stream: plain accumulation, mod 2^32)
        args.mhdr->clk_cycles += (u32)(clock64() - stamp_c0);
        args.mhdr->clk_ticks += (u32)(wall_clock64() - stamp_w0);
    }
    if (FMT == VGF_P2TR) {
        // product tree of the lanes' final products (as seq_fwd_kernel does for the denominators)
        __syncthreads();
        fe sib, pair;
        shfl_xor_fe(sib, zrun, 1);
        fe_mul(pair, zrun, sib);
        if ((tid & 1) == 0) lds_store_fe(tree, WG, WG / 2 + (tid >> 1), pair);
        __syncthreads();
#pragma unroll 1
        for (int width = WG / 4; width >= 1; width >>= 1) {
            if (tid < width) {
                const int k = width + tid;
                fe a, b, p;
                lds_load_fe(tree, WG, 2 * k, a);
                lds_load_fe(tree, WG, 2 * k + 1, b);
                fe_mul(p, a, b);
                lds_store_fe(tree, WG, k, p);
            }
            __syncthreads();
        }
        u32 *t2 = args.tree2 + (size_t)blockIdx.x * 9 * WG;
#pragma unroll
        for (int i = 0; i < 9; i++) t2[i * WG + tid] = tree[i * WG + tid];
        if (tid < 9) args.root2[(size_t)tid * args.groups + blockIdx.x] = tree[tid * WG + 1];
    }
}

// ---- stage 4 (split form): one key per lane — hashes, filter, output -------------------------------------------------
//
// The hash pair is three quarters of a key's instructions and needs 28 registers; the point arithmetic in front of it needs 128.
// In one kernel the pair therefore runs at FOUR waves per SIMD and one launch of 2^20 keys is one wave per SIMD: the device is
// only full with a dozen dispatches in flight, and a short run is mostly fill and drain.  Split, the pair runs at EIGHT waves per
// SIMD (where its issue-slot yields pay most, profiles/r04_hash_yield_ubench.jsonl) and ONE launch is sixteen waves per SIMD:
// the device is full from the first dispatch.  The price: 36 B per key through L2 / Infinity Cache between the two kernels.
// grid = (lanes / WG, 2S / hash_kpl): a lane hashes hash_kpl keys, the key steps 2j + sgn = blockIdx.y * hash_kpl + k of lane
// u = blockIdx.x * WG + tid of seq_bwd_kernel<.., SPLIT>.  hash_kpl sets how many waves per SIMD ONE launch brings (16 / hash_kpl at
// 2^20 keys): launches that fill the SIMDs' wave slots by themselves (hash_kpl = 1, 2) keep the other frames' short kernels waiting
// for slots and the command processor's pipes busy placing workgroups; 4 leaves room for a second launch and for the point
// arithmetic of the next dispatches beside it (profiles/r05_split_ab.txt).
template <int FMT, bool FULL>
struct HashWaves {   // the Base58Check encoder + DFA walk of the P2PKH matcher need 80 registers (64: 44 B of scratch)
    static constexpr int value = FULL && FMT == VGF_P2PKH ? VG_SEQ_WAVES_HASH_FULL : VG_SEQ_WAVES_HASH;
};
template <int FMT, bool FULL>
__global__ void __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(HashWaves<FMT, FULL>::value, HashWaves<FMT, FULL>::value)))
seq_hash_kernel(const SeqArgs args) {
#if VG_BASE_PRIO
    __builtin_amdgcn_s_setprio(VG_BASE_PRIO);
#endif
    static_assert(FMT == VGF_P2PKH || FMT == VGF_P2SH_P2WPKH, "compressed-key formats (P2WPKH shares P2PKH's payload)");
    extern __shared__ u32 dfa_lds[];   // FULL: the DFA blob
    const int tid = threadIdx.x;
    if (FULL) {
        for (u32 i = tid; i < args.dfa_bytes / 4; i += WG) dfa_lds[i] = args.dfa_blob[i];
        __syncthreads();
    }
    const u32 lanes = args.lanes, S = args.s, half = args.n >> 1;
    const u32 u = blockIdx.x * WG + tid;
    const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && args.mhdr != nullptr;   // shader-clock sample, as seq_bwd_kernel's
    const unsigned long long stamp_c0 = stamp ? clock64() : 0ull, stamp_w0 = stamp ? wall_clock64() : 0ull;
    const bool dump = args.dump != nullptr;
#pragma unroll 1
    for (u32 k = 0; k < args.hash_kpl; k++) {
        const u32 step = blockIdx.y * args.hash_kpl + k;
        const u32 *in = args.xs + (size_t)step * lanes + u;
        u32 xw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) xw[i] = in[(size_t)i * args.n];
        const u32 prefix = in[(size_t)8 * args.n];
        u32 pl[5];
        if (FMT == VGF_P2SH_P2WPKH) {
            u32 h[5];
            hash160_pub33_block(prefix, xw, h);
            hash160_script22_block(h, pl);
        } else {
            hash160_pub33_block(prefix, xw, pl);
        }
        const u32 j = step >> 1, sgn = step & 1u;
        const u32 index = sgn ? (half - (u + 1) * S + j) : (half + u * S + j);
        if (dump) {
            u32 *o = args.dump + (size_t)index * 5;
#pragma unroll
            for (int i = 0; i < 5; i++) o[i] = pl[i];
        } else {
            const bool hit = FULL ? dfa_match_payload_n<5, MatchFmt<FMT>::value>(dfa_lds, (int)args.fmt, pl) : filter_eval_n<5>(args.filter, pl);
            const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
            if (hit && slot < args.match_cap) {
                DevMatch *m = args.mrec + slot;
                m->index = index;
                m->reserved = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) m->payload[i] = i < 5 ? pl[i] : 0u;
            }
        }
    }
    if (stamp && tid == 0) {
        args.mhdr->clk_cycles += (u32)(clock64() - stamp_c0);
        args.mhdr->clk_ticks += (u32)(wall_clock64() - stamp_w0);
    }
}

// ---- taproot, stage C: second shared inversion walked back, x(Q), filter ---------------------------------------
//
// After seq_bwd_kernel<P2TR> (stage A) and seq_inv_kernel on root2: every lane recovers 1/(product of its 2S
// Z's) from the tree, then peels 1/Z of each key step off it in reverse order (two multiplications per key, as
// seq_bwd does for the denominators), x(Q) = X / Z^2, and the usual dump / prefilter / DFA output.
template <bool FULL>
__global__ void __launch_bounds__(WG) p2tr_finish_kernel(const SeqArgs args) {
    __shared__ u32 tree[9 * WG];
    extern __shared__ u32 dfa_lds[];
    const int tid = threadIdx.x;
    if (FULL)
        for (u32 i = tid; i < args.dfa_bytes / 4; i += WG) dfa_lds[i] = args.dfa_blob[i];
    const u32 S = args.s, lanes = args.lanes;
    const u32 u = blockIdx.x * WG + tid;
    const u32 steps = 2 * S;

    const u32 *t2 = args.tree2 + (size_t)blockIdx.x * 9 * WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tree[i * WG + tid] = t2[i * WG + tid];
    __syncthreads();
    if (tid < 9) tree[tid * WG + 1] = args.root2[(size_t)tid * args.groups + blockIdx.x];   // root^-1
    __syncthreads();
#pragma unroll 1
    for (int width = 1; width <= WG / 4; width <<= 1) {
        if (tid < width) {
            const int k = width + tid;
            fe ik, a, b, ia, ib;
            lds_load_fe(tree, WG, k, ik);
            lds_load_fe(tree, WG, 2 * k, a);
            lds_load_fe(tree, WG, 2 * k + 1, b);
            fe_mul(ia, ik, b);
            fe_mul(ib, ik, a);
            lds_store_fe(tree, WG, 2 * k, ia);
            lds_store_fe(tree, WG, 2 * k + 1, ib);
        }
        __syncthreads();
    }
    fe inv;   // 1 / (product of this lane's Z's up to the step being peeled)
    {
        fe ip, sib;
        lds_load_fe(tree, WG, WG / 2 + (tid >> 1), ip);
#pragma unroll
        for (int i = 0; i < 9; i++) sib.n[i] = args.tq[((size_t)(steps - 1) * 27 + 18 + i) * lanes + (u ^ 1u)];
        fe_mul(inv, ip, sib);
    }
    const u32 half = args.n >> 1;
    const bool dump = args.dump != nullptr;
#pragma unroll 1
    for (int step = (int)steps - 1; step >= 0; step--) {
        const u32 *in = args.tq + (size_t)step * 27 * lanes + u;
        fe X, Z, zi;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            X.n[i] = in[(size_t)i * lanes];
            Z.n[i] = in[(size_t)(9 + i) * lanes];
        }
        if (step > 0) {
            fe pprev;   // the lane's running product up to the previous step
            const u32 *prev = args.tq + (size_t)(step - 1) * 27 * lanes + u;
#pragma unroll
            for (int i = 0; i < 9; i++) pprev.n[i] = prev[(size_t)(18 + i) * lanes];
            fe_mul(zi, inv, pprev);
            fe_mul(inv, inv, Z);
        } else {
            zi = inv;
        }
        gej qq;
        qq.x = X;
        u32 xw[8], pl[8];
        taproot_affine_x(qq, zi, xw);
#pragma unroll
        for (int i = 0; i < 8; i++) pl[i] = bswap32(xw[7 - i]);   // 32 big-endian bytes in memory order
        const bool ok = args.tq_flag[(size_t)step * lanes + u] != 0;

        // stage A walks j = S-1 .. 0 and, within j, +R then -R
        const u32 j = S - 1 - ((u32)step >> 1), sgn = (u32)step & 1u;
        const u32 index = sgn ? (half - (u + 1) * S + j) : (half + u * S + j);
        if (dump) {
            u32 *o = args.dump + (size_t)index * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = ok ? pl[i] : 0u;
        } else {
            const bool hit = ok && (FULL ? dfa_match_payload_n<8>(dfa_lds, VGF_P2TR, pl) : filter_eval_n<8>(args.filter, pl));
            const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
            if (hit && slot < args.match_cap) {
                DevMatch *m = args.mrec + slot;
                m->index = index;
                m->reserved = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) m->payload[i] = pl[i];
            }
        }
    }
}

// ---- taproot behind the arbitrary-scalar path: tweak, shared inversion, output key -----------------------------------------
//
// What the reference leaves to the HOST for every key of a P2TR batch (XOnlyPublicKey::from_slice + Address::p2tr,
// src/gpu.rs:1287-1291, "CPU bound by design" src/shaders/search_p2tr.wgsl:112).  keys_bwd_kernel<P2TR> hands over the affine
// internal keys P in key order (`pts`, 64 bytes per key; y = 0 marks "no key"), then:
//   p2tr_tweak_kernel  one key per lane: t = TapTweak(x(P)), Q = lift_x(P) + t*G over the wide-window table (core/taproot.h);
//                      X(Q), Z(Q) and a validity word are parked, the Z's of the workgroup go into a product tree, its root out;
//   seq_inv_kernel     the roots of all workgroups, one per lane;
//   p2tr_out_kernel    tree down-sweep, 1/Z per lane, x(Q) = X / Z^2, the usual dump / prefilter / DFA output.
// Exactly the staging of keys_fwd / keys_bwd (round 2 finished this path inside keys_bwd_kernel with one lane inverting for its
// whole workgroup).  The sequential path keeps the tweak inside seq_bwd_kernel<P2TR> (see there): fewer instructions per key.

// Product tree over the KEYS_WG lanes of a workgroup (leaf pairs by lane shuffle), tree and root written out.
__device__ __forceinline__ void keys_tree_up(u32 *tree, const fe &z, u32 *tree_out, u32 *root, u32 groups) {
    const int tid = threadIdx.x;
    fe sib, pair;
    shfl_xor_fe(sib, z, 1);
    fe_mul(pair, z, sib);
    if ((tid & 1) == 0) lds_store_fe(tree, KEYS_WG, KEYS_WG / 2 + (tid >> 1), pair);
    __syncthreads();
#pragma unroll 1
    for (int width = KEYS_WG / 4; width >= 1; width >>= 1) {
        if (tid < width) {
            const int kk = width + tid;
            fe a, b, p;
            lds_load_fe(tree, KEYS_WG, 2 * kk, a);
            lds_load_fe(tree, KEYS_WG, 2 * kk + 1, b);
            fe_mul(p, a, b);
            lds_store_fe(tree, KEYS_WG, kk, p);
        }
        __syncthreads();
    }
    u32 *tg = tree_out + (size_t)blockIdx.x * 9 * KEYS_WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tg[i * KEYS_WG + tid] = tree[i * KEYS_WG + tid];
    if (tid < 9) root[(size_t)tid * groups + blockIdx.x] = tree[tid * KEYS_WG + 1];
}

// The way back: tree (with the inverted root) walked down in LDS; returns 1 / (z * z_sibling) of the caller's lane pair.
__device__ __forceinline__ void keys_tree_down(u32 *tree, const u32 *tree_in, const u32 *root, u32 groups, fe &ip) {
    const int tid = threadIdx.x;
    const u32 *tg = tree_in + (size_t)blockIdx.x * 9 * KEYS_WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tree[i * KEYS_WG + tid] = tg[i * KEYS_WG + tid];
    __syncthreads();
    if (tid < 9) tree[tid * KEYS_WG + 1] = root[(size_t)tid * groups + blockIdx.x];   // root^-1
    __syncthreads();
#pragma unroll 1
    for (int width = 1; width <= KEYS_WG / 4; width <<= 1) {
        if (tid < width) {
            const int kk = width + tid;
            fe ik, a, b, ia, ib;
            lds_load_fe(tree, KEYS_WG, kk, ik);
            lds_load_fe(tree, KEYS_WG, 2 * kk, a);
            lds_load_fe(tree, KEYS_WG, 2 * kk + 1, b);
            fe_mul(ia, ik, b);
            fe_mul(ib, ik, a);
            lds_store_fe(tree, KEYS_WG, 2 * kk, ia);
            lds_store_fe(tree, KEYS_WG, 2 * kk + 1, ib);
        }
        __syncthreads();
    }
    lds_load_fe(tree, KEYS_WG, KEYS_WG / 2 + (tid >> 1), ip);
}

__global__ void __launch_bounds__(KEYS_WG) __attribute__((amdgpu_waves_per_eu(3, 3))) p2tr_tweak_kernel(const KeysArgs args) {
    __shared__ u32 tree[9 * KEYS_WG];
    const u32 idx = blockIdx.x * KEYS_WG + threadIdx.x;
    const u32 lanes = args.groups * KEYS_WG;
    // the tweak needs x only: the internal key is loaded again for the addition that follows the multiplication, so that its
    // eighteen limbs are not live across the ~145-register window loop (three waves per SIMD instead of two)
    const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(args.pts + (size_t)(idx < args.n ? idx : 0) * 16);
    u32 xw[8], k[8];
#pragma unroll
    for (int q4 = 0; q4 < 2; q4++) {
        const ec_u4 t4 = e4[q4];
#pragma unroll
        for (int i = 0; i < 4; i++) xw[4 * q4 + i] = t4.v[i];
    }
    const bool okq = taproot_tweak_scalar(xw, k);
    gej tg;
    ec_mul_gen_tables(tg, k, GenTables{args.gtab, args.gtab16, args.gtab_bits});
    u32 w[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
        const ec_u4 t4 = e4[q4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[4 * q4 + i] = t4.v[i];
    }
    u32 ynz = 0;
#pragma unroll
    for (int i = 8; i < 16; i++) ynz |= w[i];
    const bool have = idx < args.n && ynz != 0;
    ge p;
    {
        fe px, py;
        fe_from_words(px, w);
        fe_from_words(py, w + 8);
        taproot_lift_even(p, px, py);
        ge g;
        ge_generator(g);      // "no key here": a harmless stand-in keeps the workgroup's product invertible; the result is discarded
#pragma unroll
        for (int i = 0; i < 9; i++) {
            p.x.n[i] = have ? p.x.n[i] : g.x.n[i];
            p.y.n[i] = have ? p.y.n[i] : g.y.n[i];
        }
    }
    gej q;
    gej_add_ge_nz(q, tg, p);                      // t*G == +/-P would need t = +/-d: negligible; Z = 0 then
    const bool zero = taproot_z_is_zero(q.z);     // t*G == -P: no address; keep the shared product invertible
    if (zero) fe_set_one(q.z);
    u32 *o = args.xyz + idx;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o[(size_t)i * lanes] = q.x.n[i];
        o[(size_t)(18 + i) * lanes] = q.z.n[i];
    }
    o[(size_t)9 * lanes] = (have && okq && !zero) ? 1u : 0u;
    keys_tree_up(tree, q.z, args.tree, args.root, args.groups);
}

template <bool FULL>
__global__ void __launch_bounds__(KEYS_WG) p2tr_out_kernel(const KeysArgs args) {
    __shared__ u32 tree[9 * KEYS_WG];
    extern __shared__ u32 dfa_lds[];    // FULL: the DFA blob
    const int tid = threadIdx.x;
    if (FULL)
        for (u32 i = tid; i < args.dfa_bytes / 4; i += KEYS_WG) dfa_lds[i] = args.dfa_blob[i];
    const u32 idx = blockIdx.x * KEYS_WG + tid;
    const u32 lanes = args.groups * KEYS_WG;
    fe ip;
    keys_tree_down(tree, args.tree, args.root, args.groups, ip);
    // 1/Z = 1/(Z * Z_sib) * Z_sib
    const u32 *in = args.xyz + idx;
    const u32 *ins = args.xyz + (idx ^ 1u);
    gej q;
    fe zs, zi;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        q.x.n[i] = in[(size_t)i * lanes];
        zs.n[i] = ins[(size_t)(18 + i) * lanes];
    }
    const bool ok = in[(size_t)9 * lanes] != 0 && idx < args.n;
    fe_mul(zi, ip, zs);
    u32 xw[8], pl[8];
    taproot_affine_x(q, zi, xw);
#pragma unroll
    for (int i = 0; i < 8; i++) pl[i] = bswap32(xw[7 - i]);   // 32 big-endian bytes in memory order
    if (idx >= args.n) return;
    if (args.dump) {
        u32 *o = args.dump + (size_t)idx * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = ok ? pl[i] : 0u;
    } else {
        const bool hit = ok && (FULL ? dfa_match_payload_n<8>(dfa_lds, VGF_P2TR, pl) : filter_eval_n<8>(args.filter, pl));
        const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
        if (hit && slot < args.match_cap) {
            DevMatch *m = args.mrec + slot;
            m->index = idx;
            m->reserved = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) m->payload[i] = pl[i];
        }
    }
}

// a: the context's output / filter / table fields, n keys whose internal keys lie in a.pts; a.xyz / a.tree / a.root: scratch
// for a.groups = ceil(n / 256) workgroups.
hipError_t launch_p2tr_tweak(const KeysArgs &a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    const bool full = a.dfa_bytes && !a.dump;
    if (full && a.dfa_bytes > DFA_MAX_BYTES) return hipErrorInvalidValue;
    if (!a.pts || !a.gtab || !a.xyz || !a.tree || !a.root || a.groups != (a.n + KEYS_WG - 1) / KEYS_WG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(p2tr_tweak_kernel, dim3(a.groups), dim3(KEYS_WG), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(seq_inv_kernel, dim3((a.groups + 63) / 64), dim3(64), 0, stream, a.root, a.groups);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (full) hipLaunchKernelGGL((p2tr_out_kernel<true>), dim3(a.groups), dim3(KEYS_WG), a.dfa_bytes, stream, a);
    else hipLaunchKernelGGL((p2tr_out_kernel<false>), dim3(a.groups), dim3(KEYS_WG), 0, stream, a);
    return hipGetLastError();
}

// ---- arbitrary scalars: full fixed-base multiplication per key ------------------------------------------------
//
// The shape of the reference's CPU path (an independent key per iteration, src/scanner.rs:151-155, full
// ec_pubkey_create each time) on the device, in the same three stages as the sequential path:
//   keys_fwd_kernel   one key per lane: k*G by fixed windows over a generator table in global memory — 24-bit windows by
//                     default (10 branch-free mixed additions over an 11.8 GB table built on the device; the 8-bit table,
//                     31 additions, when the wide one is not worth building or cannot be had): unsigned digits accumulated
//                     low to high keep the running sum below the next addend's scalar, so P = +/-Q cannot occur; "still
//                     at infinity" is a select; the Jacobian result goes to scratch (Y = 0 marks "no key"), the Z's of
//                     the workgroup into a product tree, its root out.
//   seq_inv_kernel    the roots of all workgroups, one per lane (shared with the sequential path).
//   keys_bwd_kernel   tree down-sweep, 1/Z per lane, affine + canonical coordinates, payload (on an endomorphism context:
//                     of the point's six images), filter.
// The scalars come from the frame's key buffer: uploaded (vgen_dispatch_keys) or drawn on the device by rnd_fill_kernel
// (vgen_dispatch_random).  A fused single kernel with the inversion inside (one lone wave inverting while the other three
// of its workgroup wait) measured 430 Mkeys/s; this form removes that serial section.  Issue-bound: 21 300 + 3 600
// instructions per key at the ~2.07 GHz the chip holds under this multiplier-dense code (profiles/pmc_keys.json) — ~9x the
// work of the sequential mode per key; it also serves the rare sequential batches that touch the group order.

constexpr u32 ORDER_N[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                            0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

// Scalar of lane `idx` as eight little-endian words — from the key buffer (uploaded by vgen_dispatch_keys, or filled on the
// device by rnd_fill_kernel for vgen_dispatch_random) or base + idx —; false (and k = 1, a harmless stand-in whose result
// is discarded) unless 0 < k < n (SecretKey::from_slice, src/address.rs:93) and idx < n.
__device__ __forceinline__ bool keys_load_scalar(const KeysArgs &args, u32 idx, u32 k[8]) {
    const bool in_range = idx < args.n;
    if (args.keys_be) {
        const u32 *src = reinterpret_cast<const u32 *>(args.keys_be) + (size_t)(in_range ? idx : 0) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = bswap32(src[7 - i]);
    } else {
        u64 c = idx;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += args.base[i];
            k[i] = (u32)c;
            c >>= 32;
        }
        if (c) {   // wrapped past 2^256: not a key
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = 0;
        }
    }
    u32 nz = 0;
    int cmp = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        nz |= k[i];
        const int d = (k[i] > ORDER_N[i]) - (k[i] < ORDER_N[i]);
        cmp = cmp == 0 ? d : cmp;
    }
    const bool valid = in_range && nz != 0 && cmp < 0;
    if (!valid) {
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = 0;
        k[0] = 1;
    }
    return valid;
}

// The random-key mode's scalars (the reference's rng.fill per candidate, src/scanner.rs:151-152): lane i draws candidate
// first_index + i of the counter-based stream (core/rnd.h: one SHA-256 compression) into the frame's key buffer, in the
// 32-byte big-endian layout vgen_dispatch_keys uploads — so that the multiplication kernels are the very same code for both
// (drawing the scalar inside keys_fwd_kernel kept the eight words live across its window loop and cost 15 %: 1.23 instead of
// 1.44 Gkeys/s; this fill is 700 instructions and 32 B per key, ~3 % of the dispatch).
__global__ void __launch_bounds__(256) rnd_fill_kernel(u32 *keys_be, u32 n, const RndSeed seed, u32 stream, u32 index_lo, u32 index_hi) {
    const u32 idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const u64 index = ((u64)index_hi << 32 | index_lo) + idx;
    u32 k[8];
    rnd_scalar(seed, stream, (u32)index, (u32)(index >> 32), k);
    ec_u4 *o = reinterpret_cast<ec_u4 *>(keys_be + (size_t)idx * 8);
    o[0] = ec_u4{{bswap32(k[7]), bswap32(k[6]), bswap32(k[5]), bswap32(k[4])}};
    o[1] = ec_u4{{bswap32(k[3]), bswap32(k[2]), bswap32(k[1]), bswap32(k[0])}};
}

hipError_t launch_rnd_fill(uint8_t *keys_be, u32 n, const RndSeed &seed, u32 stream, unsigned long long first_index, hipStream_t st) {
    if (!keys_be || n == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rnd_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<u32 *>(keys_be), n, seed,
                       stream, (u32)first_index, (u32)(first_index >> 32));
    return hipGetLastError();
}

__global__ void __launch_bounds__(KEYS_WG) keys_fwd_kernel(const KeysArgs args) {
    __shared__ u32 tree[9 * KEYS_WG];
    const int tid = threadIdx.x;
    const u32 idx = blockIdx.x * KEYS_WG + tid;
    const u32 lanes = args.groups * KEYS_WG;
    u32 k[8];
    const bool valid = keys_load_scalar(args, idx, k);

    gej acc;
    ec_mul_gen_tables(acc, k, GenTables{args.gtab, args.gtab16, args.gtab_bits});

    // "no key here" travels to keys_bwd_kernel as Y = 0 (all limbs): no finite point of this odd-order group has y = 0, so
    // the scalar (an upload, or a SHA-256 in the random-stream mode) is not loaded / drawn and range-checked a second time
    u32 *o = args.xyz + idx;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o[(size_t)i * lanes] = acc.x.n[i];
        o[(size_t)(9 + i) * lanes] = valid ? acc.y.n[i] : 0u;
        o[(size_t)(18 + i) * lanes] = acc.z.n[i];
    }
    // product tree of the Z's (never zero: 0 < k < n); the leaf level is exchanged by a lane shuffle
    fe sib, pair;
    shfl_xor_fe(sib, acc.z, 1);
    fe_mul(pair, acc.z, sib);
    if ((tid & 1) == 0) lds_store_fe(tree, KEYS_WG, KEYS_WG / 2 + (tid >> 1), pair);
    __syncthreads();
#pragma unroll 1
    for (int width = KEYS_WG / 4; width >= 1; width >>= 1) {
        if (tid < width) {
            const int kk = width + tid;
            fe a, b, p;
            lds_load_fe(tree, KEYS_WG, 2 * kk, a);
            lds_load_fe(tree, KEYS_WG, 2 * kk + 1, b);
            fe_mul(p, a, b);
            lds_store_fe(tree, KEYS_WG, kk, p);
        }
        __syncthreads();
    }
    u32 *tg = args.tree + (size_t)blockIdx.x * 9 * KEYS_WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tg[i * KEYS_WG + tid] = tree[i * KEYS_WG + tid];
    if (tid < 9) args.root[(size_t)tid * args.groups + blockIdx.x] = tree[tid * KEYS_WG + 1];
}

// ENDO (endomorphism contexts; every format but P2TR): the six images of every point, as seq_bwd_kernel<FMT, FULL, ENDO> tests them
// — six keys hashed for one scalar multiplication (k, lambda k, lambda^2 k and their negations; image `variant` of key i is
// reported / dumped at variant * vstride + i, vstride = the context's batch size).
template <int FMT, bool FULL, bool ENDO = false>
__global__ void __launch_bounds__(KEYS_WG) keys_bwd_kernel(const KeysArgs args) {
#if VG_BASE_PRIO
    __builtin_amdgcn_s_setprio(VG_BASE_PRIO);   // (the level the hash blocks return to; see VG_BASE_PRIO)
#endif
    __shared__ u32 tree[9 * KEYS_WG];
    __shared__ u32 ypark[ENDO && (FMT == VGF_P2PKH_UNCOMPRESSED || FMT == VGF_ETHEREUM) ? 9 * KEYS_WG : 1];   // ENDO: the point's y
    extern __shared__ u32 dfa_lds[];    // FULL: the DFA blob
    constexpr int NW = PayloadWords<FMT>::value;
    const int tid = threadIdx.x;
    if (FULL)
        for (u32 i = tid; i < args.dfa_bytes / 4; i += KEYS_WG) dfa_lds[i] = args.dfa_blob[i];
    const u32 idx = blockIdx.x * KEYS_WG + tid;
    const u32 lanes = args.groups * KEYS_WG;

    const u32 *tg = args.tree + (size_t)blockIdx.x * 9 * KEYS_WG;
#pragma unroll
    for (int i = 0; i < 9; i++) tree[i * KEYS_WG + tid] = tg[i * KEYS_WG + tid];
    __syncthreads();
    if (tid < 9) tree[tid * KEYS_WG + 1] = args.root[(size_t)tid * args.groups + blockIdx.x];   // root^-1
    __syncthreads();
#pragma unroll 1
    for (int width = 1; width <= KEYS_WG / 4; width <<= 1) {
        if (tid < width) {
            const int kk = width + tid;
            fe ik, a, b, ia, ib;
            lds_load_fe(tree, KEYS_WG, kk, ik);
            lds_load_fe(tree, KEYS_WG, 2 * kk, a);
            lds_load_fe(tree, KEYS_WG, 2 * kk + 1, b);
            fe_mul(ia, ik, b);
            fe_mul(ib, ik, a);
            lds_store_fe(tree, KEYS_WG, 2 * kk, ia);
            lds_store_fe(tree, KEYS_WG, 2 * kk + 1, ib);
        }
        __syncthreads();
    }
    // 1/Z = 1/(Z * Z_sib) * Z_sib
    fe ip, zs, zi, zi2, zi3, X, Y, x, y;
    lds_load_fe(tree, KEYS_WG, KEYS_WG / 2 + (tid >> 1), ip);
    const u32 *in = args.xyz + idx;
    const u32 *ins = args.xyz + (idx ^ 1u);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        X.n[i] = in[(size_t)i * lanes];
        Y.n[i] = in[(size_t)(9 + i) * lanes];
        zs.n[i] = ins[(size_t)(18 + i) * lanes];
    }
    u32 ynz = 0;     // keys_fwd_kernel parked Y = 0 for "no key" (scalar 0 or >= n, lane beyond n)
#pragma unroll
    for (int i = 0; i < 9; i++) ynz |= Y.n[i];
    const bool valid = ynz != 0 && idx < args.n;
    fe_mul(zi, ip, zs);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(x, X, zi2);
    fe_mul(y, Y, zi3);
    fe_canonicalize_product(x);
    fe_canonicalize_product(y);

    if (FMT == VGF_P2TR) {
        // taproot: hand the affine internal key over to p2tr_tweak_kernel (y = 0: no key here)
        if (idx >= args.n) return;
        u32 xw[8], yw[8];
        fe_to_words(x, xw);
        fe_to_words(y, yw);
        ec_u4 *o = reinterpret_cast<ec_u4 *>(args.pts + (size_t)idx * 16);
        o[0] = ec_u4{{xw[0], xw[1], xw[2], xw[3]}};
        o[1] = ec_u4{{xw[4], xw[5], xw[6], xw[7]}};
        o[2] = valid ? ec_u4{{yw[0], yw[1], yw[2], yw[3]}} : ec_u4{{0u, 0u, 0u, 0u}};
        o[3] = valid ? ec_u4{{yw[4], yw[5], yw[6], yw[7]}} : ec_u4{{0u, 0u, 0u, 0u}};
        return;
    }
    if (ENDO) {
        // (every lane has read its pair's inverse from the tree before the multiplications above; after this barrier the tree's
        //  LDS parks the x of the point in hand, as in seq_bwd_kernel)
        constexpr bool NEEDS_Y = FMT == VGF_P2PKH_UNCOMPRESSED || FMT == VGF_ETHEREUM;
        __syncthreads();
        const u32 ypar = y.n[0] & 1u;
        lds_park_fe(tree, KEYS_WG, tid, x);
        if (NEEDS_Y) lds_park_fe(ypark, KEYS_WG, tid, y);
        const bool live = valid && idx < args.n;
#pragma unroll 1
        for (u32 v = 0; v < 6; v++) {
            const u32 e = v >> 1, sneg = v & 1u;   // (x,+) (x,-) (bx,+) (bx,-) (b^2 x,+) (b^2 x,-)
            fe xe, ye;
            lds_unpark_fe(tree, KEYS_WG, tid, xe);
            if (sneg == 0 && e > 0) {
                fe beta;
                fe_set_beta(beta);
                fe_mul(xe, xe, beta);
                fe_canonicalize_product(xe);
                lds_park_fe(tree, KEYS_WG, tid, xe);
            }
            if (NEEDS_Y) {
                lds_unpark_fe(ypark, KEYS_WG, tid, ye);
                if (sneg) {
                    fe ny;
                    fe_neg(ny, ye, 1);
                    fe_normalize(ny);      // p - y, canonical (y != 0 on this curve)
                    ye = ny;
                }
            } else {
                ye.n[0] = ypar ^ sneg;     // all a compressed key reads of y
            }
            u32 ple[NW];
            (void)payload_from_point<FMT == VGF_P2TR ? VGF_P2PKH : FMT>(xe, ye, ple);
            if (idx >= args.n) continue;
            const u32 vindex = (sneg * 3u + e) * args.vstride + idx;
            if (args.dump) {
                u32 *o = args.dump + (size_t)vindex * NW;
#pragma unroll
                for (int i = 0; i < NW; i++) o[i] = live ? ple[i] : 0u;
            } else {
                const bool hit = live && (FULL ? dfa_match_payload_n<NW, MatchFmt<FMT>::value>(dfa_lds, (int)args.fmt, ple) : filter_eval_n<NW>(args.filter, ple));
                const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
                if (hit && slot < args.match_cap) {
                    DevMatch *m = args.mrec + slot;
                    m->index = vindex;
                    m->reserved = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) m->payload[i] = i < NW ? ple[i] : 0u;
                }
            }
        }
        return;
    }
    u32 pl[NW];
    const bool ok = payload_from_point<FMT == VGF_P2TR ? VGF_P2PKH : FMT>(x, y, pl) && valid;

    if (idx >= args.n) return;
    if (args.dump) {
        u32 *o = args.dump + (size_t)idx * NW;
#pragma unroll
        for (int i = 0; i < NW; i++) o[i] = ok ? pl[i] : 0u;
    } else {
        const bool hit = ok && (FULL ? dfa_match_payload_n<NW, MatchFmt<FMT>::value>(dfa_lds, (int)args.fmt, pl) : filter_eval_n<NW>(args.filter, pl));
        const u32 slot = match_slot(args.mhdr, hit, args.match_base);   // one atomic per wave
        if (hit && slot < args.match_cap) {
            DevMatch *m = args.mrec + slot;
            m->index = idx;
            m->reserved = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) m->payload[i] = i < NW ? pl[i] : 0u;
        }
    }
}

template <int FMT>
static hipError_t launch_keys_fmt(const KeysArgs &a, hipStream_t stream, hipEvent_t before_bwd) {
    const bool full = a.dfa_bytes && !a.dump;
    if (full && a.dfa_bytes > DFA_MAX_BYTES) return hipErrorInvalidValue;
    if (!a.xyz || !a.tree || !a.root || a.groups != (a.n + KEYS_WG - 1) / KEYS_WG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(keys_fwd_kernel, dim3(a.groups), dim3(KEYS_WG), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(seq_inv_kernel, dim3((a.groups + 63) / 64), dim3(64), 0, stream, a.root, a.groups);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (before_bwd && (e = hipEventRecord(before_bwd, stream)) != hipSuccess) return e;
    if (FMT == VGF_P2TR) {
        // the affine internal keys go to `pts`; tweak, second shared inversion and output are the taproot stage's
        if (!a.pts) return hipErrorInvalidValue;
        hipLaunchKernelGGL((keys_bwd_kernel<FMT, false>), dim3(a.groups), dim3(KEYS_WG), 0, stream, a);
        return hipGetLastError();   // (runtime.cpp follows with launch_p2tr_tweak over the frame's taproot scratch)
    }
    constexpr int F = FMT == VGF_P2TR ? VGF_P2PKH : FMT;   // (P2TR returned above: keeps its ENDO instantiations out of the binary)
    if (a.endo && full) hipLaunchKernelGGL((keys_bwd_kernel<F, true, true>), dim3(a.groups), dim3(KEYS_WG), a.dfa_bytes, stream, a);
    else if (a.endo) hipLaunchKernelGGL((keys_bwd_kernel<F, false, true>), dim3(a.groups), dim3(KEYS_WG), 0, stream, a);
    else if (full) hipLaunchKernelGGL((keys_bwd_kernel<FMT, true>), dim3(a.groups), dim3(KEYS_WG), a.dfa_bytes, stream, a);
    else hipLaunchKernelGGL((keys_bwd_kernel<FMT, false>), dim3(a.groups), dim3(KEYS_WG), 0, stream, a);
    return hipGetLastError();
}

// ---- wide fixed-window generator table, built on the device -----------------------------------------------------
// Entry (w, d) = d * 2^(bits w) * G for d = 1 .. 2^bits - 1: one lane per entry multiplies through the 8-bit table and
// normalises with its own inversion (~610 field multiplications per entry: 1 M entries ~3 ms of the chip, once per
// context).  Entries whose scalar would not fit 256 bits (top window) are never addressed and stay unwritten.
// Output: eight little-endian words of x, eight of y per entry (core/ec.h: ec_mul_gen_wide).
__global__ void __launch_bounds__(256) gen_table_wide_kernel(const u32 *tab8, u32 *tab, u32 bits, unsigned long long entries, unsigned long long first) {
    const unsigned long long idx = first + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= entries) return;
    const unsigned long long per = (1ull << bits) - 1ull;
    const u32 w = (u32)(idx / per);
    const unsigned long long d = idx % per + 1ull;
    const u32 bit = w * bits;
    if (bit + (64 - __clzll(d)) > 256) return;          // d * 2^bit >= 2^256: not a digit any scalar has
    u32 k[9];
#pragma unroll
    for (int i = 0; i < 9; i++) k[i] = 0;
    const u32 i0 = bit >> 5, sh = bit & 31u;
    const unsigned long long lo = d << sh;              // d < 2^24, sh < 32: fits 64 bits
    k[i0] = (u32)lo;
    if (i0 + 1 < 8) k[i0 + 1] = (u32)(lo >> 32);
    gej p;
    ec_mul_gen_w8(p, k, tab8);
    fe zi, zi2, zi3, x, y;
    fe_inv(zi, p.z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(x, p.x, zi2);
    fe_mul(y, p.y, zi3);
    fe_canonicalize_product(x);
    fe_canonicalize_product(y);
    u32 xw[8], yw[8];
    fe_to_words(x, xw);
    fe_to_words(y, yw);
    ec_u4 *o = reinterpret_cast<ec_u4 *>(tab + idx * 16ull);
    o[0] = ec_u4{{xw[0], xw[1], xw[2], xw[3]}};
    o[1] = ec_u4{{xw[4], xw[5], xw[6], xw[7]}};
    o[2] = ec_u4{{yw[0], yw[1], yw[2], yw[3]}};
    o[3] = ec_u4{{yw[4], yw[5], yw[6], yw[7]}};
}

// The same table in two levels: a digit d of `bits` bits splits into d = d_lo + 2^h d_hi (h = bits / 2), and
//   d 2^(bits w) G = d_lo 2^(h 2w) G + d_hi 2^(h (2w+1)) G = small[2w][d_lo] + small[2w+1][d_hi],
// one AFFINE addition of two entries of the h-bit table (same layout, a few thousand entries per window, built by
// gen_table_wide_kernel).  A lane makes eight consecutive digits (same d_hi) and shares one inversion among their
// additions: ~45 field multiplications per entry instead of ~610 — 20 bits in ~3 ms instead of ~40, which is what a
// cold first P2TR match waits for.  No pair is exceptional: d_lo 2^(bits w) = +/- d_hi 2^(bits w + h) has no
// solution with d_lo < 2^h.  Digits whose scalar would not fit 256 bits are skipped as in the one-level kernel.
constexpr int GT_K = 8;   // digits per lane
__global__ void __launch_bounds__(256) gen_table_combine_kernel(const u32 *small, u32 *tab, u32 bits, unsigned long long groups, unsigned long long first) {
    const unsigned long long g = first + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const u32 h = bits >> 1;
    const unsigned long long per = (1ull << bits) - 1ull, per_h = (1ull << h) - 1ull;
    const unsigned long long groups_per_window = (1ull << bits) / GT_K;
    const u32 w = (u32)(g / groups_per_window);
    const u32 d0 = (u32)(g % groups_per_window) * GT_K;     // digits d0 .. d0 + 7, same d_hi
    const u32 d_hi = d0 >> h, lo0 = d0 & (u32)per_h;
    const u32 bit = w * bits;
    const u32 *win_lo = small + (unsigned long long)(2 * w) * per_h * 16ull;
    const u32 *win_hi = small + (unsigned long long)(2 * w + 1) * per_h * 16ull;
    auto load_words = [](const u32 *ent, u32 xw[8], u32 yw[8]) {
        const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(ent);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const ec_u4 a = e4[q], b = e4[2 + q];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xw[4 * q + i] = a.v[i];
                yw[4 * q + i] = b.v[i];
            }
        }
    };
    auto valid = [&](u32 d) { return d != 0 && bit + (32u - (u32)__clz(d)) <= 256u; };
    u32 hxw[8], hyw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) hxw[i] = hyw[i] = 0;
    const bool have_hi = d_hi != 0 && valid(d0 | 1u);     // (the smallest digit of the group with this d_hi fits)
    if (have_hi) load_words(win_hi + (unsigned long long)(d_hi - 1) * 16ull, hxw, hyw);
    fe hx, hy;
    fe_from_words(hx, hxw);
    fe_from_words(hy, hyw);

    // forward: prefix products of the denominators (1 where no addition is needed)
    fe pre[GT_K];
    fe acc;
#pragma unroll
    for (int j = 0; j < GT_K; j++) {
        const u32 d = d0 + j, d_lo = lo0 + j;
        const bool add = have_hi && d_lo != 0 && valid(d);
        fe dx;
        fe_set_one(dx);
        if (add) {
            u32 xw[8], yw[8];
            load_words(win_lo + (unsigned long long)(d_lo - 1) * 16ull, xw, yw);
            fe lx;
            fe_from_words(lx, xw);
            fe_sub_n(dx, lx, hx);
        }
        if (j == 0) acc = dx;
        else fe_mul(acc, acc, dx);
        pre[j] = acc;
    }
    fe inv;
    fe_inv(inv, acc);
    // backward: each digit's inverse, the addition, the entry
#pragma unroll
    for (int j = GT_K - 1; j >= 0; j--) {
        const u32 d = d0 + j, d_lo = lo0 + j;
        const bool ok = valid(d);
        const bool add = have_hi && d_lo != 0 && ok;
        u32 xw[8], yw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) xw[i] = yw[i] = 0;
        if (ok && d_lo != 0) load_words(win_lo + (unsigned long long)(d_lo - 1) * 16ull, xw, yw);
        fe lx, ly, dx;
        fe_from_words(lx, xw);
        fe_from_words(ly, yw);
        fe_set_one(dx);
        if (add) fe_sub_n(dx, lx, hx);
        fe idx;
        if (j > 0) {
            fe_mul(idx, inv, pre[j - 1]);
            fe_mul(inv, inv, dx);
        } else {
            idx = inv;
        }
        if (!ok) continue;
        u32 oxw[8], oyw[8];
        if (add) {
            fe dy, lam, x3, t, y3;
            fe_sub_n(dy, ly, hy);
            fe_mul(lam, dy, idx);
            fe_sqr(x3, lam);
            fe_sub_n(x3, x3, lx);
            fe_sub_n(x3, x3, hx);
            fe_sub_n(t, lx, x3);
            fe_mul(y3, lam, t);
            fe_sub_n(y3, y3, ly);
            fe_canonicalize(x3);
            fe_canonicalize(y3);
            fe_to_words(x3, oxw);
            fe_to_words(y3, oyw);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                oxw[i] = d_lo != 0 ? xw[i] : hxw[i];     // d_hi == 0: the low entry itself; d_lo == 0: the high one
                oyw[i] = d_lo != 0 ? yw[i] : hyw[i];
            }
        }
        ec_u4 *o = reinterpret_cast<ec_u4 *>(tab + ((unsigned long long)w * per + (d - 1)) * 16ull);
        o[0] = ec_u4{{oxw[0], oxw[1], oxw[2], oxw[3]}};
        o[1] = ec_u4{{oxw[4], oxw[5], oxw[6], oxw[7]}};
        o[2] = ec_u4{{oyw[0], oyw[1], oyw[2], oyw[3]}};
        o[3] = ec_u4{{oyw[4], oyw[5], oyw[6], oyw[7]}};
    }
}

// ---- the signed-window tables (core/ec.h: ec_mul_gen_signed) --------------------------------------------------------------
// Same two levels.  Half-width table: for window w two half-windows of 2^h entries (h = (ST - 1) / 2):
//     small[2w][m]     = m * 2^(ST w)     * G         small[2w + 1][m] = m * 2^(ST w + h) * G        m = 1 .. 2^h
// (one lane per entry: 8-bit multiplication + own inversion); wide entry (w, m), m = 1 .. 2^(ST-1), = small[2w][m_lo] +
// small[2w+1][m_hi] with m = m_lo + 2^h m_hi — m_hi reaches 2^h for the one magnitude 2^(ST-1), which is why the half-windows
// have 2^h entries and not 2^h - 1.  Entries whose scalar would pass 2^256 are never addressed and stay unwritten — except
// the scalar 2^256 ITSELF (the top window's largest magnitude: a raw top digit of all ones plus a carry from below, e.g.
// k = 2^256 - 2^232 + 2^231 + 2^203 + 5 < n at 29 bits), which is taken mod n: 2^256 - n.
__device__ __forceinline__ bool signed_entry_scalar(u32 m, u32 bit, u32 k[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) k[i] = 0;
    const u32 len = 32u - (u32)__clz(m);
    if (bit + len <= 256u) {
        const u32 i0 = bit >> 5, sh = bit & 31u;
        const unsigned long long lo = (unsigned long long)m << sh;     // m < 2^29 (the top entry 2^28 << 31 fits 64 bits)
        k[i0] = (u32)lo;
        if (i0 + 1 < 8) k[i0 + 1] = (u32)(lo >> 32);
        return true;
    }
    if (bit + len == 257u && (m & (m - 1u)) == 0u) {   // m * 2^bit == 2^256  ->  2^256 mod n
        k[0] = 0x2FC9BEBFu; k[1] = 0x402DA173u; k[2] = 0x50B75FC4u; k[3] = 0x45512319u; k[4] = 1u;
        return true;
    }
    return false;
}

__global__ void __launch_bounds__(256) gen_small_signed_kernel(const u32 *tab8, u32 *small, u32 st, unsigned long long entries, unsigned long long first) {
    const unsigned long long idx = first + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= entries) return;
    const u32 h = (st - 1u) / 2u;
    const u32 hw = (u32)(idx >> h), m = (u32)(idx & ((1ull << h) - 1ull)) + 1u;      // half-window, magnitude 1 .. 2^h
    const u32 bit = st * (hw >> 1) + h * (hw & 1u);
    u32 k[9];
    if (!signed_entry_scalar(m, bit, k)) return;
    gej p;
    ec_mul_gen_w8(p, k, tab8);
    fe zi, zi2, zi3, x, y;
    fe_inv(zi, p.z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(x, p.x, zi2);
    fe_mul(y, p.y, zi3);
    fe_canonicalize_product(x);
    fe_canonicalize_product(y);
    u32 xw[8], yw[8];
    fe_to_words(x, xw);
    fe_to_words(y, yw);
    ec_u4 *o = reinterpret_cast<ec_u4 *>(small + idx * 16ull);
    o[0] = ec_u4{{xw[0], xw[1], xw[2], xw[3]}};
    o[1] = ec_u4{{xw[4], xw[5], xw[6], xw[7]}};
    o[2] = ec_u4{{yw[0], yw[1], yw[2], yw[3]}};
    o[3] = ec_u4{{yw[4], yw[5], yw[6], yw[7]}};
}

// groups of GT_K consecutive magnitudes m0 + 1 .. m0 + GT_K of one window (same m_hi for all but the group's last magnitude
// when that is a multiple of 2^h — handled by looking m_hi up per magnitude), one shared inversion per lane.
__global__ void __launch_bounds__(256) gen_combine_signed_kernel(const u32 *small, u32 *tab, u32 st, unsigned long long groups, unsigned long long first) {
    const unsigned long long g = first + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const u32 h = (st - 1u) / 2u, nw = ec_signed_windows(st);
    const unsigned long long per = ec_signed_per(st), per_h = 1ull << h, top_per = ec_signed_top_per(st);
    const unsigned long long full_groups = per / GT_K;                 // per window 0 .. nw-2
    u32 w;
    unsigned long long gi;
    if (g < (unsigned long long)(nw - 1u) * full_groups) {
        w = (u32)(g / full_groups);
        gi = g % full_groups;
    } else {
        w = nw - 1u;
        gi = g - (unsigned long long)(nw - 1u) * full_groups;
    }
    const unsigned long long limit = w == nw - 1u ? top_per : per;     // magnitudes 1 .. limit
    const u32 m0 = (u32)(gi * GT_K);                                   // this lane: magnitudes m0 + 1 .. m0 + GT_K
    const u32 bit = st * w;
    const u32 *win_lo = small + (unsigned long long)(2u * w) * per_h * 16ull;
    const u32 *win_hi = small + (unsigned long long)(2u * w + 1u) * per_h * 16ull;
    auto load_words = [](const u32 *ent, u32 xw[8], u32 yw[8]) {
        const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(ent);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const ec_u4 a = e4[q], b = e4[2 + q];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xw[4 * q + i] = a.v[i];
                yw[4 * q + i] = b.v[i];
            }
        }
    };
    // is magnitude m an entry of this window (its scalar m * 2^bit at most 2^256)?
    auto valid = [&](u32 m) {
        if (m == 0 || m > limit) return false;
        const u32 len = 32u - (u32)__clz(m);
        return bit + len <= 256u || (bit + len == 257u && (m & (m - 1u)) == 0u);
    };
    fe pre[GT_K];
    fe acc;
    // forward: prefix products of the denominators (1 where no addition is needed: m_lo == 0 or m_hi == 0 or not an entry)
#pragma unroll
    for (int j = 0; j < GT_K; j++) {
        const u32 m = m0 + 1u + (u32)j, m_lo = m & (u32)(per_h - 1ull), m_hi = m >> h;
        const bool add = valid(m) && m_lo != 0 && m_hi != 0;
        fe dx;
        fe_set_one(dx);
        if (add) {
            u32 xw[8], yw[8], hx[8], hy[8];
            load_words(win_lo + (unsigned long long)(m_lo - 1u) * 16ull, xw, yw);
            load_words(win_hi + (unsigned long long)(m_hi - 1u) * 16ull, hx, hy);
            fe lx, hxf;
            fe_from_words(lx, xw);
            fe_from_words(hxf, hx);
            fe_sub_n(dx, lx, hxf);
        }
        if (j == 0) acc = dx;
        else fe_mul(acc, acc, dx);
        pre[j] = acc;
    }
    fe inv;
    fe_inv(inv, acc);
#pragma unroll
    for (int j = GT_K - 1; j >= 0; j--) {
        const u32 m = m0 + 1u + (u32)j, m_lo = m & (u32)(per_h - 1ull), m_hi = m >> h;
        const bool ok = valid(m);
        const bool add = ok && m_lo != 0 && m_hi != 0;
        u32 xw[8], yw[8], hxw[8], hyw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) xw[i] = yw[i] = hxw[i] = hyw[i] = 0;
        if (ok && m_lo != 0) load_words(win_lo + (unsigned long long)(m_lo - 1u) * 16ull, xw, yw);
        if (ok && m_hi != 0) load_words(win_hi + (unsigned long long)(m_hi - 1u) * 16ull, hxw, hyw);
        fe lx, ly, hx, hy, dx;
        fe_from_words(lx, xw);
        fe_from_words(ly, yw);
        fe_from_words(hx, hxw);
        fe_from_words(hy, hyw);
        fe_set_one(dx);
        if (add) fe_sub_n(dx, lx, hx);
        fe idx;
        if (j > 0) {
            fe_mul(idx, inv, pre[j - 1]);
            fe_mul(inv, inv, dx);
        } else {
            idx = inv;
        }
        if (!ok) continue;
        u32 oxw[8], oyw[8];
        if (add) {
            fe dy, lam, x3, t, y3;
            fe_sub_n(dy, ly, hy);
            fe_mul(lam, dy, idx);
            fe_sqr(x3, lam);
            fe_sub_n(x3, x3, lx);
            fe_sub_n(x3, x3, hx);
            fe_sub_n(t, lx, x3);
            fe_mul(y3, lam, t);
            fe_sub_n(y3, y3, ly);
            fe_canonicalize(x3);
            fe_canonicalize(y3);
            fe_to_words(x3, oxw);
            fe_to_words(y3, oyw);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                oxw[i] = m_lo != 0 ? xw[i] : hxw[i];     // m_hi == 0: the low entry itself; m_lo == 0: the high one
                oyw[i] = m_lo != 0 ? yw[i] : hyw[i];
            }
        }
        const unsigned long long base = (unsigned long long)w * per;    // (every window below the top one is full)
        ec_u4 *o = reinterpret_cast<ec_u4 *>(tab + (base + (m - 1u)) * 16ull);
        o[0] = ec_u4{{oxw[0], oxw[1], oxw[2], oxw[3]}};
        o[1] = ec_u4{{oxw[4], oxw[5], oxw[6], oxw[7]}};
        o[2] = ec_u4{{oyw[0], oyw[1], oyw[2], oyw[3]}};
        o[3] = ec_u4{{oyw[4], oyw[5], oyw[6], oyw[7]}};
    }
}

// tab: the wide table (ec_table_words(bits) words); small: scratch for the half-width table (ec_table_small_words(bits) words).
// The build has two phases — 0: the half-width table, one lane per entry; 1: every wide entry as the sum of two of those, GT_K
// entries per lane — and either can be launched in SLICES of lanes [first, first + count): the runtime replaces a table in use by
// a wider one without pausing the scan, a slice riding in front of each dispatch on the frame's own stream (runtime.cpp: GtabJob).
// Phase 1 may start once every slice of phase 0 has completed.
unsigned long long gen_table_phase_lanes(u32 bits, int phase) {
    if (bits == 25 || bits == 27 || bits == 29)
        return phase == 0 ? ec_signed_small_entries(bits)
                          : (unsigned long long)(ec_signed_windows(bits) - 1u) * (ec_signed_per(bits) / GT_K) + ec_signed_top_per(bits) / GT_K;
    if (bits != 16 && bits != 20 && bits != 22 && bits != 24 && bits != 26) return 0;
    return phase == 0 ? ec_wide_entries(bits / 2) : (unsigned long long)ec_wide_windows(bits) * ((1ull << bits) / GT_K);
}

hipError_t launch_gen_table_slice(const u32 *tab8, u32 *tab, u32 *small, u32 bits, int phase, unsigned long long first,
                                  unsigned long long count, hipStream_t stream) {
    const unsigned long long total = gen_table_phase_lanes(bits, phase);
    if (total == 0 || (phase != 0 && phase != 1)) return hipErrorInvalidValue;
    if (first >= total || count == 0) return hipSuccess;
    if (count > total - first) count = total - first;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    const unsigned long long end = first + count;   // the kernels' bound: lanes beyond the slice do nothing
    const bool sgn = ec_table_signed(bits);
    if (phase == 0) {
        if (sgn) hipLaunchKernelGGL(gen_small_signed_kernel, grid, block, 0, stream, tab8, small, bits, end, first);
        else hipLaunchKernelGGL(gen_table_wide_kernel, grid, block, 0, stream, tab8, small, bits / 2, end, first);
    } else {
        if (sgn) hipLaunchKernelGGL(gen_combine_signed_kernel, grid, block, 0, stream, small, tab, bits, end, first);
        else hipLaunchKernelGGL(gen_table_combine_kernel, grid, block, 0, stream, small, tab, bits, end, first);
    }
    return hipGetLastError();
}

hipError_t launch_gen_table_wide(const u32 *tab8, u32 *tab, u32 *small, u32 bits, hipStream_t stream) {
    if (gen_table_phase_lanes(bits, 0) == 0) return hipErrorInvalidValue;
    for (int phase = 0; phase < 2; phase++) {
        const hipError_t e = launch_gen_table_slice(tab8, tab, small, bits, phase, 0, gen_table_phase_lanes(bits, phase), stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- offset table of the sequential path, built on the device -------------------------------------------------
// R_u = base + u * step, one lane per entry: up to nbits branch-free mixed additions (summand b = 2^b * step where bit b
// of u is set), then the lane's own inversion.  65 536 entries take ~0.2 ms of the chip; on the host's sixteen threads
// the same table took 7 ms of vgen_create, most of the time to a cold first match.
__global__ void __launch_bounds__(256) rtab_build_kernel(RtabArgs a) {
    const u32 u = blockIdx.x * 256 + threadIdx.x;
    if (u >= a.lanes) return;
    gej acc;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc.x.n[i] = a.base.x[i];
        acc.y.n[i] = a.base.y[i];
        acc.z.n[i] = i == 0 ? 1u : 0u;
    }
    acc.inf = 0;
#pragma unroll 1
    for (u32 b = 0; b < a.nbits; b++) {
        ge p;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            p.x.n[i] = a.pw[b].x[i];
            p.y.n[i] = a.pw[b].y[i];
        }
        gej sum;
        gej_add_ge_nz(sum, acc, p);
        const bool take = ((u >> b) & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            acc.x.n[i] = take ? sum.x.n[i] : acc.x.n[i];
            acc.y.n[i] = take ? sum.y.n[i] : acc.y.n[i];
            acc.z.n[i] = take ? sum.z.n[i] : acc.z.n[i];
        }
    }
    fe zi, zi2, zi3, x, y;
    fe_inv(zi, acc.z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(x, acc.x, zi2);
    fe_mul(y, acc.y, zi3);
    fe_canonicalize_product(x);
    fe_canonicalize_product(y);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a.rtab[(size_t)i * a.lanes + u] = x.n[i];
        a.rtab[(size_t)(9 + i) * a.lanes + u] = y.n[i];
    }
}

hipError_t launch_rtab_build(const RtabArgs &a, hipStream_t stream) {
    if (a.lanes == 0 || a.nbits > 24 || (a.nbits < 32 && (a.lanes - 1) >> a.nbits)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rtab_build_kernel, dim3((a.lanes + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---- shader-clock probe -----------------------------------------------------------------------------------
// One wave that sleeps on the SALU for `ticks` of the constant 100 MHz counter (s_memrealtime) and reports
// how far the shader-clock counter (s_memtime) moved meanwhile: the clock the CUs really ran at while the scan
// kernels were executing beside it (power management holds it below the nominal 2.4 GHz under this load).
// Every launch is time-bounded, so it always terminates.
__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long *out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w = w0;
    while (w - w0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        w = wall_clock64();
    }
    out[0] += clock64() - c0;   // launches of one probe are ordered on their stream: plain accumulation
    out[1] += w - w0;
}

// The probe is a train of 0.5 ms launches rather than one long kernel: HIP may place the probe's stream on
// a hardware queue that a frame's stream also uses, and a single long-running kernel would then hold that
// frame's kernels back for the whole window (seen: 11.9 -> 1.4 Gkeys/s); short launches let them interleave.
hipError_t launch_clock_probe(unsigned long long *out, unsigned long long ticks, hipStream_t stream) {
    const unsigned long long slice = 50000;   // 0.5 ms of the 100 MHz counter
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), stream);
    for (unsigned long long done = 0; e == hipSuccess && done < ticks; done += slice) {
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, ticks - done < slice ? ticks - done : slice);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_keys_scan(int fmt, const KeysArgs &a, hipStream_t stream, hipEvent_t before_bwd) {
    if (a.n == 0) return hipSuccess;
    switch (fmt) {
    case VGF_P2PKH:
    case VGF_P2WPKH:
        return launch_keys_fmt<VGF_P2PKH>(a, stream, before_bwd);
    case VGF_P2SH_P2WPKH:
        return launch_keys_fmt<VGF_P2SH_P2WPKH>(a, stream, before_bwd);
    case VGF_P2PKH_UNCOMPRESSED:
        return launch_keys_fmt<VGF_P2PKH_UNCOMPRESSED>(a, stream, before_bwd);
    case VGF_ETHEREUM:
        return launch_keys_fmt<VGF_ETHEREUM>(a, stream, before_bwd);
    case VGF_P2TR:
        return launch_keys_fmt<VGF_P2TR>(a, stream, before_bwd);
    default:
        return hipErrorInvalidValue;
    }
}

// ---- launch (called from runtime.cpp) ---------------------------------------------------------------------

template <int FMT>
static hipError_t launch_bwd(const SeqArgs &a, hipStream_t stream) {
    const bool full = a.dfa_bytes && !a.dump;
    if (full && a.dfa_bytes > DFA_MAX_BYTES) return hipErrorInvalidValue;
    if (FMT == VGF_P2TR) {
        // stage A (tweaked points parked) -> second root inversion -> stage C (finish + filter)
        if (!a.gtab || !a.tq || !a.tq_flag || !a.tree2 || !a.root2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((seq_bwd_kernel<FMT, false>), dim3(a.groups), dim3(WG), 0, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(seq_inv_kernel, dim3((a.groups + 63) / 64), dim3(64), 0, stream, a.root2, a.groups);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (full) hipLaunchKernelGGL((p2tr_finish_kernel<true>), dim3(a.groups), dim3(WG), a.dfa_bytes, stream, a);
        else hipLaunchKernelGGL((p2tr_finish_kernel<false>), dim3(a.groups), dim3(WG), 0, stream, a);
        return hipGetLastError();
    }
    if (a.xs && !a.endo && (FMT == VGF_P2PKH || FMT == VGF_P2SH_P2WPKH)) {
        // the split form: point arithmetic (x and the prefix byte parked per key), then one key per lane through the hashes
        constexpr int HF = FMT == VGF_P2SH_P2WPKH ? VGF_P2SH_P2WPKH : VGF_P2PKH;
        hipLaunchKernelGGL((seq_bwd_kernel<VGF_P2PKH, false, false, false, true>), dim3(a.groups), dim3(WG), 0, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (a.hash_kpl == 0 || (2 * a.s) % a.hash_kpl != 0) return hipErrorInvalidValue;
        if (full) hipLaunchKernelGGL((seq_hash_kernel<HF, true>), dim3(a.groups, 2 * a.s / a.hash_kpl), dim3(WG), a.dfa_bytes, stream, a);
        else hipLaunchKernelGGL((seq_hash_kernel<HF, false>), dim3(a.groups, 2 * a.s / a.hash_kpl), dim3(WG), 0, stream, a);
        return hipGetLastError();
    }
    if (full && a.endo && FMT != VGF_P2TR) hipLaunchKernelGGL((seq_bwd_kernel<(FMT == VGF_P2TR ? VGF_P2PKH : FMT), true, true>), dim3(a.groups), dim3(WG), a.dfa_bytes, stream, a);
    else if (full) hipLaunchKernelGGL((seq_bwd_kernel<FMT, true>), dim3(a.groups), dim3(WG), a.dfa_bytes, stream, a);
    else if (a.endo && FMT != VGF_P2TR) hipLaunchKernelGGL((seq_bwd_kernel<(FMT == VGF_P2TR ? VGF_P2PKH : FMT), false, true>), dim3(a.groups), dim3(WG), 0, stream, a);
    else if (a.lone && FMT == VGF_P2PKH) hipLaunchKernelGGL((seq_bwd_kernel<VGF_P2PKH, false, false, true>), dim3(a.groups), dim3(WG), 0, stream, a);   // (nearly) alone on the device: the twin without yields
    else hipLaunchKernelGGL((seq_bwd_kernel<FMT, false>), dim3(a.groups), dim3(WG), 0, stream, a);
    return hipGetLastError();
}


hipError_t launch_seq_fwd(const SeqArgs &a, hipStream_t stream) {
    if (a.lanes % WG != 0 || a.groups != a.lanes / WG || a.s < 2 || a.s > SEQ_MAX_S) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seq_fwd_kernel, dim3(a.groups), dim3(WG), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(seq_inv_kernel, dim3((a.groups + 63) / 64), dim3(64), 0, stream, a.root, a.groups);
    return hipGetLastError();
}

hipError_t launch_seq_bwd(int fmt, const SeqArgs &a, hipStream_t stream) {
    if (a.lanes % WG != 0 || a.groups != a.lanes / WG || a.s < 2 || a.s > SEQ_MAX_S) return hipErrorInvalidValue;
    switch (fmt) {
    case VGF_P2PKH:
    case VGF_P2WPKH:
        return launch_bwd<VGF_P2PKH>(a, stream);
    case VGF_P2SH_P2WPKH:
        return launch_bwd<VGF_P2SH_P2WPKH>(a, stream);
    case VGF_P2PKH_UNCOMPRESSED:
        return launch_bwd<VGF_P2PKH_UNCOMPRESSED>(a, stream);
    case VGF_ETHEREUM:
        return launch_bwd<VGF_ETHEREUM>(a, stream);
    case VGF_P2TR:
        return launch_bwd<VGF_P2TR>(a, stream);
    default:
        return hipErrorInvalidValue;
    }
}

}  // namespace vg
