// runtime.h — the HIP runtime object behind vgen_ctx: device binding, frames (stream + buffers +
// events), offset table, dispatch and readback.  MI355X-native stand-in for the reference's
// GpuRunner / Frame pair (src/gpu.rs:101-131): hipMalloc'ed buffers sized once, one stream per frame
// so that consecutive dispatches overlap on the device, pinned host staging for the small
// per-dispatch uploads and the rare match records, hipEvents around every launch.
#pragma once
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/vgen_hip.h"
#include "device/device_types.h"
#include "host/filter.h"
#include "host/host_ec.h"
#include "host/scalar.h"

struct vgen_ctx {
    int device = 0;
    uint32_t batch = 0, frames = 0, match_cap = 0, format = 0;
    uint32_t S = 0, lanes = 0, groups = 0;
    vg::SeqBaseCache base_cache;         // host-side incremental base points (host_ec.h)
    uint32_t payload_words = 5;
    bool timing = false;                 // VGEN_FLAG_TIMING: events around every dispatch

    uint32_t *d_rtab = nullptr;          // [18][lanes]
    uint32_t *d_gtab = nullptr;          // 8-bit fixed-window generator table (arbitrary-scalar path, P2TR), built on first use
    vg::DevFilter *d_filter = nullptr;   // current device filter program
    uint32_t *d_dfa = nullptr;           // DEVF_DFA automaton of the current filter
    uint32_t *d_chk_lut = nullptr;       // Bech32 checksum tables of the current filter (when it tests the checksum)
    bool have_filter = false;            // false = dump mode
    vg::DevFilter h_filter{};

    struct Frame {
        hipStream_t stream = nullptr;
        hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_stop = nullptr;   // before fwd / before bwd / after bwd
        uint32_t *d_dump = nullptr;
        uint8_t *d_keys = nullptr;       // explicit keys of vgen_dispatch_keys
        uint32_t *d_keys_scratch = nullptr;   // arbitrary-scalar path: Jacobian results | tree | roots (first use)
        uint32_t *d_p2tr_scratch = nullptr;   // P2TR: tweaked points | flags | second tree | second roots (first use)
        uint64_t keys_tested = 0;
        uint32_t *d_scratch = nullptr;   // pre | tree | root (device_types.h / kernels.hip)
        uint32_t match_base = 0;         // candidate counter value when the last dispatch was enqueued
        uint8_t *d_match = nullptr;      // DevMatchHeader followed by match_cap DevMatch
        uint8_t *h_match = nullptr;      // pinned mirror
        bool in_flight = false;
        bool dumped = false;             // last dispatch ran in dump mode
        vg::Scalar start{};
        bool timing_fresh = true;        // last_ms / last_total_ms already read from the events
        float last_ms = 0.f;             // dominant kernel (seq_bwd) of the last completed dispatch
        float last_total_ms = 0.f;       // whole dispatch: fwd + inv + bwd
    };
    std::vector<Frame> fr;
    uint8_t *d_slab = nullptr;                   // device memory of all frames (scratch | match ring, per frame)
    uint8_t *h_slab = nullptr;                   // pinned mirrors of the match rings
    hipStream_t probe_stream = nullptr;          // shader-clock probe (vgen_clock_probe_*)
    unsigned long long *d_probe = nullptr;
    bool probe_running = false;
    std::string err;

    int fail(int status, const std::string &msg) {
        err = msg;
        return status;
    }
};

namespace vg {

int rt_create(const vgen_params *p, vgen_ctx **out, std::string &err);
void rt_destroy(vgen_ctx *ctx);
int rt_set_filter(vgen_ctx *ctx, const vgen_filter *f);
int rt_dispatch(vgen_ctx *ctx, uint32_t frame, const uint8_t start_key_be[32]);
int rt_dispatch_keys(vgen_ctx *ctx, uint32_t frame, const uint8_t *keys_be, uint32_t n);
int rt_wait(vgen_ctx *ctx, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches,
            uint64_t *keys_tested);
int rt_read_dump(vgen_ctx *ctx, uint32_t frame, uint8_t *out, size_t out_len);
int rt_frame_times(vgen_ctx *ctx, uint32_t frame, float *kernel_ms, float *total_ms);
int rt_clock_probe_start(vgen_ctx *ctx, uint32_t duration_ms);
int rt_clock_probe_read(vgen_ctx *ctx, double *mhz);

}  // namespace vg
