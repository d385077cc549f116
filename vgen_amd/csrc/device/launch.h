// launch.h — what the host runtime sees of the kernels (no HIP device code here).
#pragma once
#include <hip/hip_runtime_api.h>

#include "device_types.h"

namespace vg {

// The sequential-range scan in its two halves (a.lanes must be a multiple of 256):
//   launch_seq_fwd  denominators, prefix products, workgroup product trees (seq_fwd_kernel) and the inverted tree roots
//                   (seq_inv_kernel behind it);
//   launch_seq_bwd  everything per key (tree walk-down, additions, hashes, filter / dump), after the first half
//                   of the same dispatch has completed (same stream, or an event between two streams).
hipError_t launch_seq_fwd(const SeqArgs &a, hipStream_t stream);
hipError_t launch_seq_bwd(int fmt, const SeqArgs &a, hipStream_t stream);

}  // namespace vg

namespace vg {
// Enqueues the arbitrary-scalar scan (one key per lane, full fixed-base multiplication, three stages like the
// sequential scan).  a.xyz / a.tree / a.root: scratch for a.groups = ceil(a.n / 256) workgroups.
// `before_bwd` (optional) is recorded between the inversion stage and the backward stage.
hipError_t launch_keys_scan(int fmt, const KeysArgs &a, hipStream_t stream, hipEvent_t before_bwd);
// The taproot stage behind either scan path (a.pts holds the affine internal keys of a.n keys): tweak multiplication and
// addition (p2tr_tweak_kernel), root inversions, output keys with dump / filter (p2tr_out_kernel).
hipError_t launch_p2tr_tweak(const KeysArgs &a, hipStream_t stream);
// Fills a key buffer (n x 32 bytes, big-endian) with candidates first_index .. first_index + n - 1 of stream `stream` under
// `seed` of the counter-based scalar stream (core/rnd.h): the random-key mode's scalars, drawn on the device.
hipError_t launch_rnd_fill(uint8_t *keys_be, uint32_t n, const RndSeed &seed, uint32_t stream, unsigned long long first_index, hipStream_t st);
// Builds the wide fixed-window generator table (bits = 16 | 20 | 22 | 24 | 26 unsigned windows, 25 | 27 | 29 signed ones;
// ec_table_words(bits) words, core/ec.h) from the 8-bit one, through a table of half the width (`small`:
// ec_table_small_words(bits) words of scratch).
hipError_t launch_gen_table_wide(const uint32_t *tab8, uint32_t *tab, uint32_t *small, uint32_t bits, hipStream_t stream);
// The same build in slices: phase 0 = the half-width table, phase 1 = the wide table from it (only after every slice of phase 0 has
// completed); lanes [first, first + count) of the phase's gen_table_phase_lanes(bits, phase).  A slice is one kernel launch.
unsigned long long gen_table_phase_lanes(uint32_t bits, int phase);
hipError_t launch_gen_table_slice(const uint32_t *tab8, uint32_t *tab, uint32_t *small, uint32_t bits, int phase, unsigned long long first,
                                  unsigned long long count, hipStream_t stream);
// Builds the sequential path's offset table on the device: rtab[(i) * lanes + u] / rtab[(9 + i) * lanes + u] = limb i of
// x / y of base + u * step (a.pw[b] = 2^b * step).  No pair (partial sum, summand) is exceptional as long as base is not a
// multiple of step and the scalars stay far below n (runtime.cpp: base = S/2, step = S).
hipError_t launch_rtab_build(const RtabArgs &a, hipStream_t stream);
// Enqueues the shader-clock probe: out[0] = shader-clock cycles, out[1] = 100 MHz ticks elapsed (>= ticks).
hipError_t launch_clock_probe(unsigned long long *out, unsigned long long ticks, hipStream_t stream);
}  // namespace vg
