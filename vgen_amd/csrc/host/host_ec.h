// host_ec.h — host-side secp256k1 helpers of libvgen_hip.so (fixed-base multiplication, batched
// affine conversion, offset tables).  Replaces the reference's host use of libsecp256k1 in
// key_to_affine (src/gpu.rs:901-910) and its on-device table build (src/shaders/init.wgsl:3-10).
#pragma once
#include <stddef.h>

#include <vector>

#include "../core/ec.h"
#include "scalar.h"

namespace vg {

// k*G as a canonical affine point; false when k == 0 (mod n) (point at infinity).
bool host_ec_mul_gen(const Scalar &k, ge &out);

// Montgomery-batched Jacobian -> affine; no input may be the point at infinity.
void host_batch_to_affine(const gej *in, ge *out, size_t n);

// out[i] = (first + i*step) * G for i in [0, count): one mixed addition per entry plus one shared
// inversion per chunk.  first >= 1, step >= 1, and first + count*step must stay far below n.
void host_build_stride_table(uint64_t first, uint64_t step, uint32_t count, std::vector<ge> &out);

}  // namespace vg

namespace vg {

// Per-context cache for the per-dispatch uniform points Q_j = (kb + j)*G, j < S.  A scan walks its
// base scalar forward by a constant stride, so after the first dispatch the S points are advanced by
// ONE shared-inversion batch of affine additions with the cached stride point instead of a fresh
// fixed-base multiplication — and, once the stride has repeated, eight dispatches' worth of points at a time
// (the Fermat inversion is most of the host's per-dispatch cost).
struct SeqBaseCache {
    static constexpr uint32_t LOOK = 8;   // dispatches computed ahead once the stride has repeated
    bool valid = false;
    uint32_t S = 0;
    Scalar kb{};
    ge q[32];
    bool dvalid = false;
    uint64_t delta = 0;
    ge dpt;
    // look-ahead: the point sets of kb + (i+1)*delta, i < ahead_n, all from ONE shared inversion
    // (LOOK*S affine additions q[j] + (i+1)*delta*G); ahead_pos = next one to hand out
    bool mvalid = false;
    ge mult[LOOK];        // (i+1) * delta * G
    uint32_t ahead_n = 0, ahead_pos = 0;
    ge ahead[LOOK][32];
};

// out[j] = (kb + j)*G for j < S.  false if any point is the point at infinity (kb + j == 0 mod n).
bool host_seq_points(SeqBaseCache &cache, const Scalar &kb, uint32_t S, ge *out);

}  // namespace vg

namespace vg {
// The 4-bit fixed-window generator table as the device wants it: [64][15][18] limbs
// (x limbs 0..8 then y limbs 0..8 of d * 16^w * G, d = 1..15).
void host_gen_table_limbs(std::vector<uint32_t> &out);
// The same generator multiples for 8-bit windows: [32][255][20] words (ec_mul_gen_w8, core/ec.h).
void host_gen_table8_limbs(std::vector<uint32_t> &out);
}  // namespace vg
