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
