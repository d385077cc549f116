// launch.h — what the host runtime sees of the kernels (no HIP device code here).
#pragma once
#include <hip/hip_runtime_api.h>

#include "device_types.h"

namespace vg {

// Enqueues the sequential-range scan on `stream`.  a.lanes must be a multiple of 256.
// `before_bwd` (optional) is recorded between the inversion stage and the backward stage.
hipError_t launch_seq_scan(int fmt, const SeqArgs &a, hipStream_t stream, hipEvent_t before_bwd);

}  // namespace vg

namespace vg {
// Enqueues the arbitrary-scalar scan (one key per lane, full fixed-base multiplication, three stages like the
// sequential scan).  a.xyz / a.tree / a.root: scratch for a.groups = ceil(a.n / 256) workgroups.
// `before_bwd` (optional) is recorded between the inversion stage and the backward stage.
hipError_t launch_keys_scan(int fmt, const KeysArgs &a, hipStream_t stream, hipEvent_t before_bwd);
// Enqueues the shader-clock probe: out[0] = shader-clock cycles, out[1] = 100 MHz ticks elapsed (>= ticks).
hipError_t launch_clock_probe(unsigned long long *out, unsigned long long ticks, hipStream_t stream);
}  // namespace vg
