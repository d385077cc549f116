// runtime.h — the HIP runtime object behind vgen_ctx: device binding, frames (buffers + events + stream), offset
// table, dispatch and readback.  MI355X-native stand-in for the reference's GpuRunner / Frame pair
// (src/gpu.rs:101-131): hipMalloc'ed buffers sized once, pinned host staging for the rare match records (and the dumps
// of dump mode).  One launch of the per-key kernel is only one wave per SIMD and a dispatch is a chain of dependent
// launches, so the device is filled by the frames in flight: every frame drives its dispatches through a private
// stream that OWNS a hardware queue (see create_stream() in runtime.cpp — no GPU_MAX_HW_QUEUES needed).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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
    bool lone_variant = true;   // dispatches issued while at most lone_max_others other frames are in flight launch seq_bwd_kernel<.., LONE>
                                // (VGEN_LONE_VARIANT=0 turns that off)
    uint32_t lone_max_others = 1;
    uint32_t hash_kpl = 4;      // split form: keys per lane of seq_hash_kernel (16 / hash_kpl waves per SIMD per launch at 2^20 keys)
    bool split = false;         // VGEN_SPLIT=1 (A/B only): compressed-key formats run point arithmetic and hashes as two kernels
    vg::SeqBaseCache base_cache;         // host-side incremental base points (host_ec.h)
    uint32_t payload_words = 5;
    bool timing = false;                 // VGEN_FLAG_TIMING: events around every dispatch
    bool endo = false;                   // VGEN_FLAG_ENDO (any format but P2TR): six keys per point where the kernels support it

    uint32_t *d_rtab = nullptr;          // [18][lanes]
    uint32_t *d_gtab = nullptr;          // 8-bit fixed-window generator table (arbitrary-scalar path, P2TR), built on first use
    uint32_t *d_gtab16 = nullptr;        // wide fixed-window generator table (gtab_bits bits): built on the device from d_gtab at first use and
                                         // SHARED by the process's contexts on this device (runtime.cpp: GtabCache; a reference, not owned)
    uint32_t gtab_bits = 0;              // width of d_gtab16
    uint32_t gtab_bits_wanted = 0;       // width asked for (see wanted_table_bits in runtime.cpp) once a dispatch wanted a wide table
    uint32_t gtab_param = 0;             // vgen_params.table_bits: the caller's choice for this context (0 = automatic)
    uint32_t gtab_env = 0;               // VGEN_GTAB_BITS as it stood at vgen_create: test override of everything else (0 = unset)
    uint32_t gtab_bits_pref = 0;         // the scan loop's choice for this scan (rt_prefer_table_bits): wider tables pay off on long scans
    uint32_t gtab_bits_cap = 0;          // ... and its cap (vgen_scan_config.table_bits_max; 0 = none)
    bool gtab_wide_failed = false;       // no wide table could be had (or the 8-bit one was asked for): stay on the 8-bit one
    std::string gtab_note;               // why the table in use is narrower than the one asked for (vgen_get_resources)
    // A table replaced by a wider one stays allocated while dispatches that read it may be in flight; it is let go at the next
    // moment the context has nothing in flight (hipFree synchronises the device: never beside a running scan).
    uint32_t *gtab_old = nullptr;
    // The wider table on its way (runtime.cpp: "replacing the table in use"): its two buffers are allocated by a thread of
    // their own (hipMalloc of 138 GB takes ~0.5 s), then built in SLICES that ride in front of the context's dispatches on
    // the frames' own streams — no extra stream (all twelve hardware queues belong to the frames), no pause.
    struct GtabJob {
        uint32_t bits = 0;
        std::thread alloc;
        std::atomic<int> alloc_state{0};     // 0 running, 1 done, 2 failed
        uint32_t *wide = nullptr, *small = nullptr;
        std::string why;
        int phase = 0;                       // 0: half-width table, 1: the wide table from it, 2: complete
        unsigned long long next = 0;         // lanes of the phase already issued
        uint32_t pending = 0;                // frames in flight that carry a slice of the current phase
        uint64_t slices = 0;
    };
    std::unique_ptr<GtabJob> gtab_job;
    uint32_t gtab_job_failed_bits = 0;   // a width whose background build failed: not tried again by this context
    uint32_t gtab_waiting_bits = 0;      // another context of this process is building this width for the device: adopt it when it is there
    uint64_t mem_budget = 0;             // vgen_params.device_mem_budget_bytes (0 = automatic)
    uint64_t frames_bytes = 0;           // device memory fixed at vgen_create (scratch slab, match rings, offset table, filter program)
    uint64_t mode_bytes = 0;             // device memory made at first use (dump buffers, arbitrary-scalar slab, 8-bit table, automaton, LUT)
    uint64_t pinned_bytes = 0;           // page-locked host memory (match-ring and dump mirrors)
    bool trace_create = false;           // VGEN_TRACE_CREATE at vgen_create
    vg::DevFilter *d_filter = nullptr;   // current device filter program
    uint32_t *d_dfa = nullptr;           // DEVF_DFA automaton of the current filter
    uint32_t *d_chk_lut = nullptr;       // Bech32 checksum tables of the current filter (when it tests the checksum)
    bool have_filter = false;            // false = dump mode
    vg::DevFilter h_filter{};

    struct Frame {
        hipStream_t s = nullptr;         // this frame's stream (owned by the context): carries the whole dispatch chain
        hipEvent_t ev_done = nullptr;    // dispatch complete incl. its copies: what vgen_wait waits on when streams share queues, no timing
        hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_stop = nullptr;   // VGEN_FLAG_TIMING: before fwd / before bwd / after bwd
        uint32_t *d_dump = nullptr;      // dump mode: slice of d_dump_slab
        uint8_t *h_dump = nullptr;       // dump mode: pinned mirror, filled by the dispatch's own async copy
        uint8_t *d_keys = nullptr;       // explicit keys of vgen_dispatch_keys (slice of d_keys_slab)
        uint32_t *d_keys_scratch = nullptr;   // arbitrary-scalar path: Jacobian results | tree | roots (slice of d_keys_slab)
        uint32_t *d_p2tr_scratch = nullptr;   // P2TR, sequential path: tweaked points | flags | second tree | second roots (slice of d_slab)
        uint32_t *d_keys_p2tr = nullptr;      // P2TR, arbitrary-scalar path: internal keys | X, Z | tree | roots of the taproot stage (slice of d_keys_slab)
        uint64_t keys_tested = 0;
        uint64_t dump_slots = 0;         // payload slots of the last dispatch's dump: batch, or 6 x batch on an endomorphism context
        uint32_t *d_scratch = nullptr;   // pre | tree | root | arrive (device_types.h / kernels.hip)
        uint32_t match_base = 0;         // candidate counter value when the last dispatch was enqueued
        uint8_t *d_match = nullptr;      // DevMatchHeader followed by match_cap DevMatch
        uint8_t *h_match = nullptr;      // pinned mirror
        bool in_flight = false;
        bool dumped = false;             // last dispatch ran in dump mode
        bool endo_applied = false;       // last dispatch tested the six images of every point (keys_tested = 6 x batch)
        vg::Scalar start{};
        bool timing_fresh = true;        // last_ms / last_total_ms already read from the events
        float last_ms = 0.f;             // dominant kernel (seq_bwd) of the last completed dispatch
        float last_total_ms = 0.f;       // whole dispatch: fwd + inv + bwd
        bool carries_slice = false;      // the dispatch in flight has a slice of the table build in front of it (GtabJob::pending)
        uint32_t clk_cycles_seen = 0, clk_ticks_seen = 0;   // match-header clock sums at the last vgen_wait
        uint32_t last_clk_cycles = 0, last_clk_ticks = 0;   // ... and what the last dispatch added to them
    };
    std::vector<Frame> fr;
    std::vector<uint64_t> sort_scratch, sort_scratch2;   // rt_wait: (index, slot) words of the records being put in index order
    // One stream per frame, created at first use (a hardware queue each, ~8 ms) or, once a scan has asked for them
    // (rt_prepare_streams), by a helper thread while the scan runs on the frames it already has.
    std::vector<hipStream_t> streams;            // [frames]
    std::mutex stream_mu;                        // guards the slots and the claims below
    std::condition_variable stream_cv;
    std::vector<char> claimed;                   // slot is being created by somebody: wait for it instead of creating another
    std::thread stream_maker;
    bool maker_started = false;
    std::atomic<bool> maker_cancel{false};
    uint32_t cu_count = 0;
    uint32_t hw_queues = 4;                      // GPU_MAX_HW_QUEUES in effect when the context was created (queues per priority level)
    uint32_t prio_levels = 1;                    // stream priority levels the runtime reports (3 on ROCm 7.2) ...
    int prio_least = 0, prio_greatest = 0;       // ... and their range (hipDeviceGetStreamPriorityRange)
    uint8_t *d_slab = nullptr;                   // device memory of all frames (scratch [| P2TR scratch], per frame)
    uint8_t *d_match_slab = nullptr;             // match rings of all frames (rt_set_match_cap)
    uint8_t *h_slab = nullptr;                   // pinned mirrors of the match rings
    uint8_t *d_dump_slab = nullptr;              // dump mode: payload buffers of frames 0 and 1 (first vgen_set_filter(NULL))
    uint8_t *h_dump_slab = nullptr;              // ... and their pinned mirrors
    uint8_t *d_dump_slab2 = nullptr, *h_dump_slab2 = nullptr;   // frames 2 .. dump_frames-1: made when one of them first dumps
    uint32_t dump_frames = 0;                    // frames that have a dump buffer (all of them unless that would pin > ~1 GiB)
    uint8_t *d_keys_slab = nullptr;              // arbitrary-scalar path: keys + scratch of all frames (first use)
    hipStream_t probe_stream = nullptr;          // shader-clock probe (vgen_clock_probe_*)
    unsigned long long *d_probe = nullptr;
    bool probe_running = false;
    std::string err;

    // Fault injection exists only in the test build of the library (-DVGEN_TEST_HOOKS: tests/native/libvgen_hip_hooks.so,
    // tests/native/vgen_hip_hooks.h); the shipped libvgen_hip.so has neither the field nor the entry point.
#ifdef VGEN_TEST_HOOKS
    uint64_t fail_after = UINT64_MAX;            // vgen_debug_fail_after: dispatches still accepted
    // true once the injected fault has struck: every later dispatch fails
    bool injected_fault() {
        if (fail_after == UINT64_MAX) return false;
        if (fail_after == 0) return true;
        fail_after--;
        return false;
    }
#else
    static constexpr bool injected_fault() { return false; }
#endif

    int fail(int status, const std::string &msg) {
        err = msg;
        return status;
    }
};

namespace vg {

// More frames than hardware queues the runtime provides (GPU_MAX_HW_QUEUES per priority level): streams then share queues.
inline bool rt_oversubscribed(const vgen_ctx *c) { return c->frames > c->prio_levels * c->hw_queues; }

// Everything the C ABI (cabi.cpp) and the scan loop (scanner.cpp) need of the device goes through these functions; the
// two files make no HIP call of their own — which is also what lets the host-side sanitizer builds link them against a
// CPU stand-in of this interface (tests/native/fake_rt.cpp).
int rt_device_count(int *n, std::string &err);
int rt_device_name(int device, std::string &name, std::string &err);
int rt_create(const vgen_params *p, vgen_ctx **out, std::string &err);
// Starts (once) a helper thread that creates the frame streams no frame has used yet; rt_frame_ready tells without
// blocking whether `frame` could be dispatched to without creating a stream first.  Used by the scan loop to grow its
// pipeline as queues become available instead of stalling ~8 ms per frame.  false: not supported for this stream kind.
bool rt_prepare_streams(vgen_ctx *ctx);
bool rt_frame_ready(vgen_ctx *ctx, uint32_t frame);
void rt_destroy(vgen_ctx *ctx);
int rt_set_filter(vgen_ctx *ctx, const vgen_filter *f);
int rt_set_match_cap(vgen_ctx *ctx, uint32_t cap);
int rt_dispatch(vgen_ctx *ctx, uint32_t frame, const uint8_t start_key_be[32]);
int rt_dispatch_keys(vgen_ctx *ctx, uint32_t frame, const uint8_t *keys_be, uint32_t n);
int rt_dispatch_random(vgen_ctx *ctx, uint32_t frame, const RndSeed &seed, uint32_t stream, uint64_t first_index);
int rt_wait(vgen_ctx *ctx, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches,
            uint64_t *keys_tested);
int rt_read_dump(vgen_ctx *ctx, uint32_t frame, uint8_t *out, size_t out_len);
int rt_dump_view(vgen_ctx *ctx, uint32_t frame, const uint8_t **ptr, size_t *len);
int rt_frame_times(vgen_ctx *ctx, uint32_t frame, float *kernel_ms, float *total_ms);
int rt_frame_clock(vgen_ctx *ctx, uint32_t frame, uint32_t *cycles, uint32_t *ticks);
int rt_clock_probe_start(vgen_ctx *ctx, uint32_t duration_ms);
int rt_clock_probe_read(vgen_ctx *ctx, double *mhz);
// The generator-table width the next scalar-multiplication dispatches should use when VGEN_GTAB_BITS does not say (0 = the default, 24).
// A wider table costs more to make (24 bits: 11.8 GB, ~30 ms; 27 signed: 21.5 GB, ~60 ms; 29 signed: 138 GB, 0.7 - 2.3 s) and saves additions on
// every key after (10 / 9 / 8 per multiplication: +5 % / +12.5 %): the scan loop asks for what the expected length of the scan pays for.
// `cap` (vgen_scan_config.table_bits_max; 0 = none) bounds it by additions per multiplication.  A wider table than the one in use is
// built in the background while the dispatches go on (runtime.cpp) and taken into use when complete: nothing waits for it.
void rt_prefer_table_bits(vgen_ctx *ctx, uint32_t bits, uint32_t cap = 0);
int rt_get_memory(const vgen_ctx *ctx, vgen_memory_info *out);
int rt_get_resources(const vgen_ctx *ctx, uint32_t *dump_frames, uint32_t *table_bits, uint32_t *table_bits_wanted, std::string *note);

}  // namespace vg
