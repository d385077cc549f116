// runtime.cpp — see runtime.h.
#include "runtime.h"

#include <chrono>

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <mutex>

#include "device/launch.h"
#include "host/host_ec.h"

namespace vg {

namespace {

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return (ctx)->fail(VGEN_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

constexpr uint32_t FIRST_COPY = 256;   // match records fetched together with the header

size_t match_bytes(uint32_t n) { return sizeof(DevMatchHeader) + (size_t)n * sizeof(DevMatch); }

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    long x = strtol(v, nullptr, 10);
    return x > 0 ? (uint32_t)x : dflt;
}

// The compressed-key formats run the per-key stage split in two (kernels.hip: seq_bwd_kernel<.., SPLIT> -> seq_hash_kernel): x and the
// prefix byte of every key wait in the frame's scratch between them.
bool split_format(const vgen_ctx *c) {
    return c->split && (c->format == VGF_P2PKH || c->format == VGF_P2WPKH || c->format == VGF_P2SH_P2WPKH);
}

// pre [S][9][lanes] | tree [groups][9][WG] | root [9][groups] | split form: xs [9][batch]
size_t scratch_words(const vgen_ctx *c) {
    return (size_t)c->S * 9 * c->lanes + (size_t)c->groups * 9 * SEQ_WG + (size_t)9 * c->groups + (split_format(c) ? (size_t)9 * c->batch : 0);
}

// P2TR: tq [2S][27][lanes] | tq_flag [2S][lanes] | tree2 [groups][9][WG] | root2 [9][groups]
size_t p2tr_words(const vgen_ctx *c) {
    if (c->format != VGF_P2TR) return 0;
    return (size_t)2 * c->S * 27 * c->lanes + (size_t)2 * c->S * c->lanes + (size_t)c->groups * 9 * SEQ_WG + (size_t)9 * c->groups;
}

// P2TR behind the arbitrary-scalar path (one key per lane, G = ceil(batch / 256) workgroups), part of the keys slab:
//   pts [batch][16] | xz [27][G * 256] | tree [G][9][256] | root [9][G]
size_t p2tr_groups(const vgen_ctx *c) { return ((size_t)c->batch + KEYS_WG - 1) / KEYS_WG; }
size_t p2tr_stage_words(const vgen_ctx *c) {
    if (c->format != VGF_P2TR) return 0;
    const size_t g = p2tr_groups(c);
    return (size_t)c->batch * 16 + (size_t)27 * g * KEYS_WG + g * 9 * KEYS_WG + (size_t)9 * g;
}

void fe_canon_neg(fe &r, const fe &a) {
    fe_neg(r, a, 1);
    fe_normalize(r);
}

size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

void release_gtab(vgen_ctx *c);   // (defined with the generator-table cache below)
bool mem_allows(vgen_ctx *c, uint64_t extra, bool table, bool by_name, std::string *why);   // (the memory policy, same place)

// Frame i's stream, created on first use (a stream costs ~5-8 ms: a scan's first dispatches should be running while
// the later streams are still being set up).
//
// Which streams own a hardware queue (measured with tools/queue_probe.hip, profiles/r02_queue_probe.txt): ordinary
// streams share the runtime's pool of GPU_MAX_HW_QUEUES (default 4) queues — the 5th, 9th, 13th stream each add a
// full serial round —, but the runtime keeps one such pool PER PRIORITY LEVEL.  The context therefore spreads its
// frames' streams over the levels hipDeviceGetStreamPriorityRange reports (three on ROCm 7.2: twelve streams own
// twelve queues with no environment variable and whether or not the host initialised HIP first); on a runtime with
// fewer levels the surplus streams share queues, which vgen_get_topology reports as `oversubscribed`.
// [CU-masked streams (hipExtStreamCreateWithCUMask) also get a queue each and measured 2-3 % faster at 12-20 frames,
// but their teardown is broken on ROCm 7.2 — see EXPERIMENTS.md "Two events of round 2" — and they are not part of the library.]

// Creates the stream of frame i (no locking here).
int create_stream(vgen_ctx *c, uint32_t i, hipStream_t *out, std::string &err) {
    *out = nullptr;
    int prio = 0;
    if (c->prio_levels >= 3) {
        // streams 0-3 take the normal level, 4-7 the next, 8-11 the third
        static const int order[3] = {0, -1, 1};
        prio = order[(i / c->hw_queues) % 3];
        if (prio < c->prio_greatest || prio > c->prio_least) prio = 0;
    } else if (c->prio_levels == 2) {
        prio = (i / c->hw_queues) % 2 ? c->prio_greatest : c->prio_least;
    }
    const hipError_t e = hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
    if (e != hipSuccess) {
        err = std::string("stream creation: ") + hipGetErrorString(e);
        return VGEN_E_HIP;
    }
    return VGEN_OK;
}

// The stream of frame i, created now if nobody has yet; if somebody (the helper thread) is creating it, waits for that.
int frame_stream(vgen_ctx *c, uint32_t i, hipStream_t *out) {
    std::unique_lock<std::mutex> lk(c->stream_mu);
    c->stream_cv.wait(lk, [&]() { return c->streams[i] || !c->claimed[i]; });
    if (!c->streams[i]) {
        c->claimed[i] = 1;
        lk.unlock();
        hipStream_t st = nullptr;
        std::string err;
        const int rc = create_stream(c, i, &st, err);
        lk.lock();
        c->claimed[i] = 0;
        c->streams[i] = st;
        c->stream_cv.notify_all();
        if (rc != VGEN_OK) return c->fail(rc, err);
    }
    *out = c->streams[i];
    return VGEN_OK;
}

}  // namespace

// Contexts whose helper thread may still be running.  A process that exits without vgen_destroy (an error path of a host
// application) must not reach the HIP runtime's teardown with such a thread inside hipStreamCreate: an atexit handler,
// registered at the first helper's start (i.e. after the runtime registered its own, so it runs before it), stops them.
struct MakerRegistry {
    std::mutex mu;
    std::vector<vgen_ctx *> live;
};
static MakerRegistry &maker_registry() {
    static MakerRegistry *r = new MakerRegistry();
    return *r;
}
static void stop_stream_maker(vgen_ctx *c) {
    c->maker_cancel.store(true);
    if (c->stream_maker.joinable()) c->stream_maker.join();   // at most one stream creation away
}
static void stop_all_stream_makers() {
    MakerRegistry &r = maker_registry();
    std::lock_guard<std::mutex> g(r.mu);
    for (vgen_ctx *c : r.live) stop_stream_maker(c);
    r.live.clear();
}

bool rt_prepare_streams(vgen_ctx *c) {
    std::lock_guard<std::mutex> g(c->stream_mu);
    if (c->maker_started) return true;
    c->maker_started = true;
    {
        static std::once_flag once;
        std::call_once(once, []() { atexit(stop_all_stream_makers); });
        MakerRegistry &r = maker_registry();
        std::lock_guard<std::mutex> rg(r.mu);
        r.live.push_back(c);
    }
    c->stream_maker = std::thread([c]() {
        if (hipSetDevice(c->device) != hipSuccess) return;
        for (uint32_t i = 0; i < c->streams.size(); i++) {
            if (c->maker_cancel.load()) return;
            {
                std::lock_guard<std::mutex> lk(c->stream_mu);
                if (c->streams[i] || c->claimed[i]) continue;
                c->claimed[i] = 1;
            }
            hipStream_t st = nullptr;
            std::string err;
            (void)create_stream(c, i, &st, err);   // on failure the slot stays empty: its first user retries and reports
            {
                std::lock_guard<std::mutex> lk(c->stream_mu);
                c->claimed[i] = 0;
                c->streams[i] = st;
            }
            c->stream_cv.notify_all();
        }
    });
    return true;
}

bool rt_frame_ready(vgen_ctx *c, uint32_t frame) {
    if (frame >= c->frames) return false;
    if (c->fr[frame].s) return true;
    std::lock_guard<std::mutex> g(c->stream_mu);
    return c->streams[frame] != nullptr;
}

namespace {

// Blocking upload that never touches the null stream (whose queue would otherwise be one more next to the frames'
// own, and which the non-blocking frame streams do not synchronise with anyway).
int upload(vgen_ctx *c, void *dst, const void *src, size_t bytes) {
    hipStream_t st;
    if (int rc = frame_stream(c, 0, &st)) return rc;
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    return VGEN_OK;
}

}  // namespace

int rt_device_count(int *n, std::string &err) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return e == hipErrorNoDevice ? VGEN_OK : VGEN_E_HIP;
    }
    *n = c;
    return VGEN_OK;
}

int rt_device_name(int device, std::string &name, std::string &err) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return VGEN_E_NODEVICE;
    }
    name = std::string(prop.name) + " (" + prop.gcnArchName + ", " + std::to_string(prop.multiProcessorCount) + " CUs)";
    return VGEN_OK;
}

int rt_create(const vgen_params *p_in, vgen_ctx **out, std::string &err) {
    // ABI 4's parameter block, or ABI 3's 28 bytes (no table_bits, no budget: both automatic)
    constexpr uint32_t PARAMS_ABI3 = 28;
    if (!p_in || !out || (p_in->struct_size != sizeof(vgen_params) && p_in->struct_size != PARAMS_ABI3)) {
        err = "vgen_create: bad parameter block";
        return VGEN_E_INVALID;
    }
    vgen_params p_full;
    memset(&p_full, 0, sizeof p_full);
    memcpy(&p_full, p_in, p_in->struct_size);
    const vgen_params *p = &p_full;
    if (p->table_bits && !(p->table_bits == 8 || p->table_bits == 16 || p->table_bits == 20 || p->table_bits == 22 || p->table_bits == 24 ||
                           p->table_bits == 26 || p->table_bits == 25 || p->table_bits == 27 || p->table_bits == 29)) {
        err = "vgen_params.table_bits must be 0 (automatic), 8, 16, 20, 22, 24 or 26 (unsigned windows) or 25, 27 or 29 (signed windows)";
        return VGEN_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        err = "no HIP device available (libvgen_hip has no CPU fallback)";
        return VGEN_E_NODEVICE;
    }
    if (p->device < 0 || p->device >= ndev) {
        err = "device index out of range";
        return VGEN_E_NODEVICE;
    }
    if (p->format > VGF_ETHEREUM) {
        err = "unknown address format";
        return VGEN_E_INVALID;
    }
    const bool trace = getenv("VGEN_TRACE_CREATE") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tlast = now();
    auto lap = [&](const char *what) { if (trace) { double t = now(); fprintf(stderr, "[vgen_create] %-28s %.2f ms\n", what, t - tlast); tlast = t; } };
    vgen_ctx *c = new vgen_ctx();
    c->trace_create = trace;
    c->device = p->device;
    c->mem_budget = p->device_mem_budget_bytes;
    c->gtab_param = p->table_bits;
    c->gtab_env = env_u32("VGEN_GTAB_BITS", 0);   // test override, read once: the dispatch path never looks at the environment
    c->batch = p->batch_size ? p->batch_size : (1u << 20);
    c->frames = p->frames ? p->frames : 12;   // the engine's own optimum (the reference's wgpu runner keeps 2, src/gpu.rs:399)
    c->match_cap = p->match_cap ? p->match_cap : 4096;
    c->format = p->format;
    c->payload_words = p->format == VGF_P2TR ? 8 : 5;
    c->timing = (p->flags & VGEN_FLAG_TIMING) != 0;
    c->endo = (p->flags & VGEN_FLAG_ENDO) != 0 && p->format != VGF_P2TR;
    c->S = env_u32("VGEN_SEQ_S", 8);
    c->hash_kpl = env_u32("VGEN_HASH_KPL", 4);   // (A/B: tools/split_ab.sh)
    if (const char *v = getenv("VGEN_SPLIT")) c->split = v[0] == '1';   // A/B switch (profiles/r05_occupancy_ab.txt: measured, not shipped)
    {   // "0": one-frame contexts launch the steady-state kernel too (counter passes: tools/pmc_valu.sh)
        const char *v = getenv("VGEN_LONE_VARIANT");
        c->lone_variant = !(v && v[0] == '0');
        c->lone_max_others = env_u32("VGEN_LONE_MAX_OTHERS", 1);   // A/B override (tools/frames_lone_ab.sh); 0 is a value here
        if (const char *w = getenv("VGEN_LONE_MAX_OTHERS")) if (w[0] == '0') c->lone_max_others = 0;
    }
    auto bail = [&](int st, const std::string &m) {
        err = m;
        rt_destroy(c);
        return st;
    };
    if (c->frames > 20)
        return bail(VGEN_E_INVALID, "frames must be <= 20 (one stream per frame: beyond ~22 busy queues per device the throughput "
                                    "collapses to ~1 Gkeys/s)");
    c->hw_queues = env_u32("GPU_MAX_HW_QUEUES", 4);   // the HIP runtime's own setting: queues per priority level
    if (c->S < 2 || c->S > SEQ_MAX_S || (c->S & (c->S - 1))) return bail(VGEN_E_INVALID, "VGEN_SEQ_S must be a power of two in [2, 16]");
    if (c->hash_kpl == 0 || (2 * c->S) % c->hash_kpl != 0) c->hash_kpl = 1;
    if (c->batch % 8192 != 0 || c->batch % (2 * SEQ_WG * c->S) != 0 || c->batch < 8192)
        return bail(VGEN_E_INVALID, "batch_size must be a multiple of 8192 (and of 512*S)");
    // the largest dispatch the kernels are laid out and tested for: 32-bit indices into 6 x batch payload slots, 24 bits of
    // lane number in the offset-table build, ~75 B of scratch per key and frame
    if (c->batch > VGEN_MAX_BATCH) return bail(VGEN_E_INVALID, "batch_size must not exceed 16777216 (2^24)");
    if (c->match_cap < FIRST_COPY) c->match_cap = FIRST_COPY;
    c->lanes = c->batch / (2 * c->S);
    c->groups = c->lanes / SEQ_WG;
    c->streams.assign(c->frames, nullptr);
    c->claimed.assign(c->frames, 0);

    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return bail(VGEN_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    {
        int cus = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device)) != hipSuccess || cus <= 0)
            return bail(VGEN_E_HIP, std::string("hipDeviceGetAttribute: ") + hipGetErrorString(e));
        c->cu_count = (uint32_t)cus;
        // stream priority levels of this runtime: one pool of hardware queues each (see create_stream)
        int least = 0, greatest = 0;
        if ((e = hipDeviceGetStreamPriorityRange(&least, &greatest)) != hipSuccess)
            return bail(VGEN_E_HIP, std::string("hipDeviceGetStreamPriorityRange: ") + hipGetErrorString(e));
        c->prio_least = least;
        c->prio_greatest = greatest;
        c->prio_levels = (uint32_t)std::max(1, least - greatest + 1);
    }

    // One device slab and one pinned slab for all frames (sixteen frames as separate allocations cost ~85 ms of
    // vgen_create, i.e. of the time to a cold first match); slices are 256-byte aligned.  Everything a dispatch
    // of this context's format touches is allocated here, nothing on the dispatch path.
    lap("checks + hipSetDevice");
    c->fr.resize(c->frames);
    const size_t scratch_b = up256(scratch_words(c) * sizeof(uint32_t));
    const size_t p2tr_b = up256(p2tr_words(c) * sizeof(uint32_t));
    const size_t frame_b = scratch_b + p2tr_b;
    {
        // what the frames take is a function of batch_size x frames (x the format), as in the reference (src/gpu.rs:391-402): known
        // before anything is allocated, and refused here when the caller's budget does not cover it
        const uint64_t fixed = (uint64_t)frame_b * c->frames + sizeof(DevFilter) + (uint64_t)18 * c->lanes * sizeof(uint32_t) +
                               (uint64_t)up256(match_bytes(c->match_cap)) * c->frames;
        if (c->mem_budget && fixed > c->mem_budget)
            return bail(VGEN_E_NOMEM, "device_mem_budget_bytes " + std::to_string(c->mem_budget) + " does not cover the frames: " + std::to_string(fixed) +
                                          " bytes for " + std::to_string(c->frames) + " frames of " + std::to_string(c->batch) + " keys");
    }
    if ((e = hipMalloc((void **)&c->d_slab, frame_b * c->frames)) != hipSuccess)
        return bail(VGEN_E_NOMEM, std::string("frame allocation: ") + hipGetErrorString(e));
    c->frames_bytes += (uint64_t)frame_b * c->frames;
    for (uint32_t i = 0; i < c->frames; i++) {
        vgen_ctx::Frame &f = c->fr[i];
        f.d_scratch = reinterpret_cast<uint32_t *>(c->d_slab + frame_b * i);
        if (p2tr_b) f.d_p2tr_scratch = reinterpret_cast<uint32_t *>(c->d_slab + frame_b * i + scratch_b);
    }
    e = hipMalloc((void **)&c->d_filter, sizeof(DevFilter));
    if (e != hipSuccess) return bail(VGEN_E_NOMEM, std::string("hipMalloc(filter): ") + hipGetErrorString(e));
    c->frames_bytes += sizeof(DevFilter);

    lap("hipMalloc slab + filter");
    // offset table R_u = (u*S + S/2) * G, limb-major ([18][lanes]) for coalesced reads — built on the device
    // (rtab_build_kernel): the host only supplies (S/2) G and the doublings 2^b S G.  [The host build of the same table,
    // host_build_stride_table on sixteen threads, was 7 of the 10 ms of vgen_create.]
    RtabArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.lanes = c->lanes;
    while (ra.nbits < 24 && ((uint64_t)1 << ra.nbits) < c->lanes) ra.nbits++;
    {
        auto point = [&](uint64_t k, DevAffine &o) -> bool {
            Scalar s{};
            s.w[0] = (uint32_t)k;
            s.w[1] = (uint32_t)(k >> 32);
            ge g;
            if (!host_ec_mul_gen(s, g)) return false;
            for (int i = 0; i < 9; i++) {
                o.x[i] = g.x.n[i];
                o.y[i] = g.y.n[i];
            }
            return true;
        };
        bool ok = point(c->S / 2, ra.base);
        for (uint32_t bit = 0; ok && bit < ra.nbits; bit++) ok = point((uint64_t)c->S << bit, ra.pw[bit]);
        if (!ok) return bail(VGEN_E_INVALID, "offset table: point at infinity");
    }
    lap("host: base + doublings");
    e = hipMalloc((void **)&c->d_rtab, (size_t)18 * c->lanes * sizeof(uint32_t));
    if (e != hipSuccess) return bail(VGEN_E_NOMEM, std::string("hipMalloc(rtab): ") + hipGetErrorString(e));
    c->frames_bytes += (uint64_t)18 * c->lanes * sizeof(uint32_t);
    ra.rtab = c->d_rtab;
    // The scratch needs no initialisation (the memset only makes a first read of never-written padding deterministic);
    // cleared and built on frame 0's stream and waited for: the other frames' streams do not synchronise with it, so
    // nothing may be pending when vgen_create returns.
    hipStream_t st0 = nullptr;
    lap("hipMalloc rtab");
    if (frame_stream(c, 0, &st0) != VGEN_OK) return bail(VGEN_E_HIP, c->err);
    lap("stream 0");
    if ((e = hipMemsetAsync(c->d_slab, 0, frame_b * c->frames, st0)) != hipSuccess ||
        (e = launch_rtab_build(ra, st0)) != hipSuccess ||
        (e = hipStreamSynchronize(st0)) != hipSuccess)
        return bail(VGEN_E_HIP, std::string("frame setup: ") + hipGetErrorString(e));
    lap("memset + table kernel + sync");
    if (int rc = rt_set_match_cap(c, c->match_cap)) return bail(rc, c->err);
    lap("match rings (device + pinned)");
    *out = c;
    return VGEN_OK;
}

// (Re)allocates the match rings of all frames for `cap` records each: at vgen_create, and when a scan meets a pattern
// too permissive for the current rings (scanner.cpp grows them instead of falling back to filtering every key on the
// host).  Only between dispatches.  The rings' monotonic counters restart at zero.
int rt_set_match_cap(vgen_ctx *c, uint32_t cap) {
    HIP_TRY(c, hipSetDevice(c->device));
    for (auto &f : c->fr)
        if (f.in_flight) return c->fail(VGEN_E_STATE, "vgen_set_match_cap while a dispatch is in flight");
    if (cap < FIRST_COPY) cap = FIRST_COPY;
    // a dispatch cannot report more candidates than it tests keys: batch, or six images of each on an ENDO context
    const uint64_t most = (uint64_t)c->batch * (c->endo ? 6 : 1);
    if (cap > most) cap = (uint32_t)std::min<uint64_t>(most, 0xFFFFFFFFu);
    if (c->d_match_slab && cap == c->match_cap) return VGEN_OK;
    const size_t match_b = up256(match_bytes(cap));
    const uint64_t old_b = c->d_match_slab ? (uint64_t)up256(match_bytes(c->match_cap)) * c->frames : 0;
    if (c->mem_budget && c->frames_bytes - old_b + c->mode_bytes + (uint64_t)match_b * c->frames > c->mem_budget)   // (the tables are not in the way of a ring)
        return c->fail(VGEN_E_NOMEM, "match rings of " + std::to_string(cap) + " records per frame pass device_mem_budget_bytes");
    uint8_t *d = nullptr, *h = nullptr;
    if (hipMalloc((void **)&d, match_b * c->frames) != hipSuccess) return c->fail(VGEN_E_NOMEM, "match ring allocation failed");
    if (hipHostMalloc((void **)&h, match_b * c->frames, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(d);
        return c->fail(VGEN_E_NOMEM, "match ring allocation failed (pinned host memory)");
    }
    // from here on every failure releases the two new buffers: the context keeps its old rings
    auto drop = [&](int rc) {
        (void)hipFree(d);
        (void)hipHostFree(h);
        return rc;
    };
    hipStream_t st0 = nullptr;
    if (int rc = frame_stream(c, 0, &st0)) return drop(rc);
    hipError_t e;
    if ((e = hipMemsetAsync(d, 0, match_b * c->frames, st0)) != hipSuccess || (e = hipStreamSynchronize(st0)) != hipSuccess)
        return drop(c->fail(VGEN_E_HIP, std::string("match ring setup: ") + hipGetErrorString(e)));
    if (c->d_match_slab) (void)hipFree(c->d_match_slab);
    if (c->h_slab) (void)hipHostFree(c->h_slab);
    c->d_match_slab = d;
    c->h_slab = h;
    c->match_cap = cap;
    c->frames_bytes += (uint64_t)match_b * c->frames - old_b;
    c->pinned_bytes += (uint64_t)match_b * c->frames - old_b;
    for (uint32_t i = 0; i < c->frames; i++) {
        vgen_ctx::Frame &f = c->fr[i];
        f.d_match = d + match_b * i;
        f.h_match = h + match_b * i;
        f.match_base = 0;
        f.clk_cycles_seen = f.clk_ticks_seen = 0;
    }
    return VGEN_OK;
}

void rt_destroy(vgen_ctx *c) {
    if (!c) return;
    {
        MakerRegistry &r = maker_registry();
        std::lock_guard<std::mutex> g(r.mu);
        r.live.erase(std::remove(r.live.begin(), r.live.end(), c), r.live.end());
    }
    stop_stream_maker(c);
    (void)hipSetDevice(c->device);
    for (auto &st : c->streams)
        if (st) (void)hipStreamSynchronize(st);
    for (auto &f : c->fr) {
        for (hipEvent_t ev : {f.ev_done, f.ev_start, f.ev_mid, f.ev_stop})
            if (ev) (void)hipEventDestroy(ev);
    }
    for (auto &st : c->streams)
        if (st) (void)hipStreamDestroy(st);
    if (c->probe_stream) {
        (void)hipStreamSynchronize(c->probe_stream);
        (void)hipStreamDestroy(c->probe_stream);
    }
    if (c->d_probe) (void)hipFree(c->d_probe);
    if (c->d_slab) (void)hipFree(c->d_slab);      // scratch + match rings of all frames
    if (c->d_match_slab) (void)hipFree(c->d_match_slab);
    if (c->h_slab) (void)hipHostFree(c->h_slab);
    if (c->d_dump_slab) (void)hipFree(c->d_dump_slab);
    if (c->h_dump_slab) (void)hipHostFree(c->h_dump_slab);
    if (c->d_dump_slab2) (void)hipFree(c->d_dump_slab2);
    if (c->h_dump_slab2) (void)hipHostFree(c->h_dump_slab2);
    if (c->d_keys_slab) (void)hipFree(c->d_keys_slab);
    if (c->d_rtab) (void)hipFree(c->d_rtab);
    if (c->d_gtab) (void)hipFree(c->d_gtab);
    release_gtab(c);                              // the wide table is shared per device: freed with its last user
    if (c->d_chk_lut) (void)hipFree(c->d_chk_lut);
    if (c->d_dfa) (void)hipFree(c->d_dfa);
    if (c->d_filter) (void)hipFree(c->d_filter);
    delete c;
}

// Shader-clock probe: a single sleeping wave on its own stream, beside whatever the frames are running.
int rt_clock_probe_start(vgen_ctx *c, uint32_t duration_ms) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->probe_running) return c->fail(VGEN_E_STATE, "a clock probe is already running");
    if (duration_ms == 0 || duration_ms > 10000) return c->fail(VGEN_E_INVALID, "probe duration must be 1..10000 ms");
    if (!c->probe_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
    if (!c->d_probe) HIP_TRY(c, hipMalloc((void **)&c->d_probe, 2 * sizeof(unsigned long long)));
    HIP_TRY(c, launch_clock_probe(c->d_probe, (unsigned long long)duration_ms * 100000ull, c->probe_stream));   // 100 MHz ticks
    c->probe_running = true;
    return VGEN_OK;
}

int rt_clock_probe_read(vgen_ctx *c, double *mhz) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->probe_running) return c->fail(VGEN_E_STATE, "no clock probe was started");
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(v, c->d_probe, sizeof v, hipMemcpyDeviceToHost, c->probe_stream));
    HIP_TRY(c, hipStreamSynchronize(c->probe_stream));
    c->probe_running = false;
    if (v[1] == 0) return c->fail(VGEN_E_HIP, "clock probe returned no ticks");
    *mhz = (double)v[0] / (double)v[1] * 100.0;
    return VGEN_OK;
}

namespace {

// Dump mode's buffers: the payloads on the device and their pinned mirrors, which every dump-mode dispatch fills with its
// own asynchronous copy (vgen_read_dump / vgen_dump_view then need no device call).  Allocated when dump mode is first
// selected — never while a dispatch of this context is in flight — for the first two frames only: pinning memory is slow
// (~0.15 ms per MiB), and a scan that wants the first match of a pattern every key satisfies (the reference's default
// `range --puzzle N`, src/lib.rs:519) never uses a third.  The remaining frames get theirs, in one more piece, when one of
// them is first dispatched in dump mode (ensure_dump_frame).
int alloc_dump_piece(vgen_ctx *c, uint32_t first, uint32_t n, uint8_t **d_out, uint8_t **h_out) {
    const size_t per = up256((size_t)c->batch * (c->endo ? 6 : 1) * c->payload_words * sizeof(uint32_t));
    uint8_t *d = nullptr, *h = nullptr;
    std::string refused;
    if (!mem_allows(c, (uint64_t)per * n, false, false, &refused)) return c->fail(VGEN_E_NOMEM, "dump buffers: " + refused);
    if (hipMalloc((void **)&d, per * n) != hipSuccess) return c->fail(VGEN_E_NOMEM, "dump buffer allocation failed");
    if (hipHostMalloc((void **)&h, per * n, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(d);
        return c->fail(VGEN_E_NOMEM, "dump buffer allocation failed (pinned host memory)");
    }
    c->mode_bytes += (uint64_t)per * n;
    c->pinned_bytes += (uint64_t)per * n;
    *d_out = d;
    *h_out = h;
    for (uint32_t i = 0; i < n; i++) {
        c->fr[first + i].d_dump = reinterpret_cast<uint32_t *>(d + per * i);
        c->fr[first + i].h_dump = h + per * i;
    }
    return VGEN_OK;
}

// Frames that dump mode serves.  Pinned host memory is the scarce part (an ENDO context at the CLI's defaults would pin
// 1.5 GB, 24 GB at 2^24 keys per dispatch): at most ~1 GiB worth of frames, never fewer than two — the host-side filter
// behind a dump runs at a few Mkeys/s, so two dumps in flight already keep it fed.  Reported by vgen_get_resources.
uint32_t dump_frames_for(const vgen_ctx *c) {
    const size_t per = up256((size_t)c->batch * (c->endo ? 6 : 1) * c->payload_words * sizeof(uint32_t));
    const size_t budget = (size_t)1 << 30;
    return (uint32_t)std::min<size_t>(c->frames, std::max<size_t>(2, budget / per));
}

int ensure_dump_slab(vgen_ctx *c) {
    if (c->d_dump_slab) return VGEN_OK;
    const uint32_t n = dump_frames_for(c);
    if (int rc = alloc_dump_piece(c, 0, std::min<uint32_t>(n, 2), &c->d_dump_slab, &c->h_dump_slab)) return rc;
    c->dump_frames = n;
    return VGEN_OK;
}

// The dump buffer of `frame`, on the dispatch path: frames 0 and 1 have theirs, the others' piece is made at first need.
int ensure_dump_frame(vgen_ctx *c, uint32_t frame) {
    if (int rc = ensure_dump_slab(c)) return rc;   // (a context that never called vgen_set_filter)
    if (frame >= c->dump_frames)
        return c->fail(VGEN_E_STATE, "dump mode serves frames 0.." + std::to_string(c->dump_frames - 1) + " of this context (pinned-memory budget; vgen_get_resources reports the limit)");
    if (c->fr[frame].d_dump) return VGEN_OK;
    return alloc_dump_piece(c, 2, c->dump_frames - 2, &c->d_dump_slab2, &c->h_dump_slab2);
}

bool dump_mode(const vgen_ctx *c) { return !c->have_filter || c->h_filter.kind == DEVF_HOST_ALL; }

}  // namespace

int rt_set_filter(vgen_ctx *c, const vgen_filter *f) {
    HIP_TRY(c, hipSetDevice(c->device));
    for (auto &fr : c->fr)
        if (fr.in_flight) return c->fail(VGEN_E_STATE, "vgen_set_filter while a dispatch is in flight");
    if (!f) {
        c->have_filter = false;
        return ensure_dump_slab(c);
    }
    if (f->format != c->format) return c->fail(VGEN_E_INVALID, "filter was compiled for another address format");
    c->h_filter = f->dev;
    if (f->dev.chk_lut) {   // Bech32 checksum tables: upload and point the device copy at them
        if (!c->d_chk_lut) {
            HIP_TRY(c, hipMalloc((void **)&c->d_chk_lut, 32 * 256 * sizeof(uint32_t)));
            c->mode_bytes += 32 * 256 * sizeof(uint32_t);
        }
        if (int rc = upload(c, c->d_chk_lut, f->chk_lut.data(), f->chk_lut.size() * sizeof(uint32_t))) return rc;
        c->h_filter.chk_lut = c->d_chk_lut;
    }
    if (f->dev.kind == DEVF_DFA) {   // the pattern's automaton: upload, point the device copy at it
        if (!c->d_dfa) {
            HIP_TRY(c, hipMalloc((void **)&c->d_dfa, 48 * 1024));
            c->mode_bytes += 48 * 1024;
        }
        if (int rc = upload(c, c->d_dfa, f->dfa_blob.data(), f->dfa_blob.size() * 4)) return rc;
        c->h_filter.dfa_blob = c->d_dfa;
    }
    // (uploads are waited for: the other frames' streams do not synchronise with the one that carried them)
    if (int rc = upload(c, c->d_filter, &c->h_filter, sizeof(DevFilter))) return rc;
    c->have_filter = true;
    if (dump_mode(c)) return ensure_dump_slab(c);
    return VGEN_OK;
}

namespace {

// The frame's stream and events, created on first use.
int ensure_frame(vgen_ctx *c, uint32_t frame) {
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.s) return VGEN_OK;
    if (int rc = frame_stream(c, frame, &f.s)) return rc;
    HIP_TRY(c, hipEventCreateWithFlags(&f.ev_done, hipEventDisableTiming));
    if (c->timing) {
        HIP_TRY(c, hipEventCreate(&f.ev_start));
        HIP_TRY(c, hipEventCreate(&f.ev_mid));
        HIP_TRY(c, hipEventCreate(&f.ev_stop));
    }
    return VGEN_OK;
}

// How vgen_wait waits.  A frame whose stream owns its hardware queue synchronises that STREAM: the HIP runtime then
// retires the stream's finished commands as the loop goes.  Waiting on an event instead leaves them to the next
// device-wide synchronisation, which a host that calls hipDeviceSynchronize / torch.cuda.synchronize() after a scan
// then pays for in one piece (measured: 0.2-0.4 ms after twenty dispatches, against 10 us).  Frames whose streams share
// hardware queues (more frames than priority levels x GPU_MAX_HW_QUEUES: more than twelve by default) wait on the
// event recorded behind their own dispatch: a stream synchronisation there also waits for the queue's other streams
// (16 frames: 10.9 instead of 12.0 Gkeys/s).
inline bool frame_owns_stream(const vgen_ctx *c) { return !rt_oversubscribed(c); }

int wait_done(vgen_ctx *c, vgen_ctx::Frame &f) {
    if (frame_owns_stream(c)) HIP_TRY(c, hipStreamSynchronize(f.s));
    else HIP_TRY(c, hipEventSynchronize(f.ev_done));
    return VGEN_OK;
}

// What follows the last kernel of a dispatch on the frame's stream: the copy of the results and the event
// vgen_wait waits on.
int finish_dispatch(vgen_ctx *c, vgen_ctx::Frame &f, bool dump, uint64_t keys, bool endo) {
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_stop, f.s));
    // payload slots of the dump: batch, or — six images per key, image v of key i at v * batch + i — 6 x batch
    f.dump_slots = (uint64_t)c->batch * (endo ? 6 : 1);
    if (dump)
        HIP_TRY(c, hipMemcpyAsync(f.h_dump, f.d_dump, (size_t)f.dump_slots * c->payload_words * sizeof(uint32_t), hipMemcpyDeviceToHost, f.s));
    else
        HIP_TRY(c, hipMemcpyAsync(f.h_match, f.d_match, match_bytes(FIRST_COPY), hipMemcpyDeviceToHost, f.s));
    if (!frame_owns_stream(c)) HIP_TRY(c, hipEventRecord(f.ev_done, f.s));
    f.in_flight = true;
    f.dumped = dump;
    f.keys_tested = keys;
    f.endo_applied = endo;
    return VGEN_OK;
}

// ---- generator tables: who decides how much device memory they take, and how a table in use is replaced ---------------------------
//
// Wide generator tables are shared by the contexts of a process that sit on the same device (tests, multi-tenant hosts,
// several scans in one host application): 11.8 GB at the default width is paid once per device, not once per context.
// Reference-counted; the last context to go frees the table.
struct GtabShared {
    int device;
    uint32_t bits;
    uint32_t *wide, *small;    // small: the half-width table the wide one was combined from (kept: freeing it would synchronise the device)
    uint32_t refs;
};
struct GtabCache {
    std::mutex mu;
    std::vector<GtabShared> tabs;
    std::vector<std::pair<int, uint32_t>> building;   // (device, bits) some context of the process is building in the background
};
GtabCache &gtab_cache() {
    static GtabCache *g = new GtabCache();   // (never destroyed: contexts may outlive static teardown order)
    return *g;
}

uint64_t gtab_bytes(uint32_t bits) { return bits <= 8 ? 0 : ((uint64_t)ec_table_words(bits) + ec_table_small_words(bits)) * sizeof(uint32_t); }
bool gtab_width_ok(uint32_t b) { return b == 8 || b == 16 || b == 20 || b == 22 || b == 24 || b == 26 || b == 25 || b == 27 || b == 29; }
// additions per multiplication of a width: what "wider" means across signed and unsigned tables
uint32_t gtab_additions(uint32_t b) { return b <= 8 ? 31u : ec_table_signed(b) ? ec_signed_windows(b) - 1u : ec_wide_windows(b) - 1u; }
// every width, by (additions per multiplication, bytes): 29s (8; 138 GB) | 27s, 26 (9; 21.5 / 43 GB) | 25s, 24 (10; 5.9 / 11.8 GB) | 22 | 20 | 16
const uint32_t GTAB_WIDTHS[] = {29, 27, 26, 25, 24, 22, 20, 16};

// Drops one reference on a shared table; the last one frees it (hipFree synchronises the device: callers pick their moment).
void gtab_unref(int device, uint32_t *wide) {
    if (!wide) return;
    GtabCache &g = gtab_cache();
    std::lock_guard<std::mutex> lk(g.mu);
    for (size_t i = 0; i < g.tabs.size(); i++) {
        GtabShared &t = g.tabs[i];
        if (t.device != device || t.wide != wide) continue;
        if (--t.refs == 0) {
            (void)hipFree(t.wide);
            (void)hipFree(t.small);
            g.tabs.erase(g.tabs.begin() + (long)i);
        }
        return;
    }
}

uint32_t gtab_bits_of(int device, const uint32_t *wide) {
    if (!wide) return 0;
    GtabCache &g = gtab_cache();
    std::lock_guard<std::mutex> lk(g.mu);
    for (GtabShared &t : g.tabs)
        if (t.device == device && t.wide == wide) return t.bits;
    return 0;
}

bool gtab_adopt_shared_peek(int device, uint32_t bits) {
    GtabCache &g = gtab_cache();
    std::lock_guard<std::mutex> lk(g.mu);
    for (GtabShared &t : g.tabs)
        if (t.device == device && t.bits == bits) return true;
    return false;
}

// Takes a reference on the device's table of `bits` bits if some context of the process has built it.
uint32_t *gtab_adopt_shared(int device, uint32_t bits) {
    GtabCache &g = gtab_cache();
    std::lock_guard<std::mutex> lk(g.mu);
    for (GtabShared &t : g.tabs)
        if (t.device == device && t.bits == bits) {
            t.refs++;
            return t.wide;
        }
    return nullptr;
}

// Joins the allocating thread of a background build and gives back whatever it holds (vgen_destroy; a failed or abandoned build).
void gtab_job_drop(vgen_ctx *c) {
    if (!c->gtab_job) return;
    vgen_ctx::GtabJob &j = *c->gtab_job;
    if (j.alloc.joinable()) j.alloc.join();
    if (j.wide) (void)hipFree(j.wide);
    if (j.small) (void)hipFree(j.small);
    {
        GtabCache &g = gtab_cache();
        std::lock_guard<std::mutex> lk(g.mu);
        auto it = std::find(g.building.begin(), g.building.end(), std::make_pair(c->device, j.bits));
        if (it != g.building.end()) g.building.erase(it);
    }
    c->gtab_job.reset();
}

// Drops the context's references on its wide tables (rt_destroy).
void release_gtab(vgen_ctx *c) {
    gtab_job_drop(c);
    gtab_unref(c->device, c->gtab_old);
    c->gtab_old = nullptr;
    gtab_unref(c->device, c->d_gtab16);
    c->d_gtab16 = nullptr;
    c->gtab_bits = 0;
}

// The retired table goes when nothing of this context is in flight (no launched kernel can hold its address, and the
// device-wide synchronisation inside hipFree costs nothing then).
void gtab_release_old_if_idle(vgen_ctx *c) {
    if (!c->gtab_old) return;
    for (auto &f : c->fr)
        if (f.in_flight) return;
    gtab_unref(c->device, c->gtab_old);
    c->gtab_old = nullptr;
}

uint64_t gtab_held_bytes(const vgen_ctx *c) {
    uint64_t b = 0;
    if (c->d_gtab16) b += gtab_bytes(c->gtab_bits);
    if (c->gtab_old) b += gtab_bytes(gtab_bits_of(c->device, c->gtab_old));
    if (c->gtab_job) b += gtab_bytes(c->gtab_job->bits);
    return b;
}

// May this context take `extra` more bytes of device memory?  With a budget (vgen_params.device_mem_budget_bytes): everything it holds
// plus `extra` stays within it.  Without one, generator tables chosen AUTOMATICALLY (table = true, nobody asked for that width by
// name) must fit half of what the device has free right now — the rule that keeps a scan from taking 138 GB of a device another
// tenant is using.  `why` says which rule refused.
bool mem_allows(vgen_ctx *c, uint64_t extra, bool table, bool by_name, std::string *why) {
    if (c->mem_budget) {
        const uint64_t held = c->frames_bytes + c->mode_bytes + gtab_held_bytes(c);
        if (held + extra <= c->mem_budget) return true;
        if (why) *why = "device_mem_budget_bytes " + std::to_string(c->mem_budget) + " < " + std::to_string(held) + " held + " + std::to_string(extra);
        return false;
    }
    if (!table || by_name) return true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    if (extra <= free_b / 2) return true;
    if (why) *why = std::to_string(extra >> 20) + " MiB is more than half of the " + std::to_string(free_b >> 20) + " MiB free on the device";
    return false;
}

// The width the context should be using: the test override, else the caller's choice for the context, else the running scan's
// (bounded by its cap), else 24 bits.  by_name: somebody asked for exactly this width.
uint32_t wanted_table_bits(const vgen_ctx *c, bool *by_name) {
    if (by_name) *by_name = c->gtab_env || c->gtab_param;
    if (c->gtab_env) return c->gtab_env;
    if (c->gtab_param) return c->gtab_param;
    uint32_t want = c->gtab_bits_pref ? c->gtab_bits_pref : 24;
    if (c->gtab_bits_cap && gtab_additions(want) < gtab_additions(c->gtab_bits_cap)) want = c->gtab_bits_cap;
    return want;
}

#ifdef VGEN_TEST_HOOKS
// fault injection (test build only): widths >= VGEN_DEBUG_GTAB_FAIL behave as if their allocation had failed
uint32_t gtab_hook_fail_from() { return env_u32("VGEN_DEBUG_GTAB_FAIL", 0); }
#else
constexpr uint32_t gtab_hook_fail_from() { return 0; }
#endif

// First wide table of a context, built here and now on frame 0's stream (24 bits: ~20 ms): `first`, or — when that cannot be
// allocated, fails to build or is refused by the memory policy — the next width in order of (additions, bytes).  Under the
// cache lock, so that two contexts of a device do not build the same table side by side.
void gtab_build_first(vgen_ctx *c, uint32_t first, bool by_name, const std::function<void(const char *)> &lap) {
    GtabCache &g = gtab_cache();
    std::lock_guard<std::mutex> lk(g.mu);
    std::string why;
    std::vector<uint32_t> order{first};
    for (uint32_t b : GTAB_WIDTHS)
        if (b != first && gtab_additions(b) >= gtab_additions(first)) order.push_back(b);
    const uint32_t hook_fail_from = gtab_hook_fail_from();
    for (uint32_t bits : order) {
        bool shared = false;   // somebody on this device has it already
        for (GtabShared &t : g.tabs)
            if (t.device == c->device && t.bits == bits) {
                t.refs++;
                c->d_gtab16 = t.wide;
                c->gtab_bits = bits;
                shared = true;
                break;
            }
        if (shared) break;
        std::string refused;
        if (!mem_allows(c, gtab_bytes(bits), true, by_name && bits == first, &refused)) {
            if (why.empty()) why = refused + " at " + std::to_string(bits) + " bits";
            continue;
        }
        uint32_t *wide_tab = nullptr, *small = nullptr;   // small: the half-width table the wide one is combined from
        hipError_t e = hook_fail_from && bits >= hook_fail_from ? hipErrorOutOfMemory
                                                                : hipMalloc((void **)&wide_tab, (size_t)ec_table_words(bits) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&small, (size_t)ec_table_small_words(bits) * sizeof(uint32_t));
        lap("hipMalloc wide + scratch");
        hipStream_t st0 = nullptr;
        if (e == hipSuccess && frame_stream(c, 0, &st0) != VGEN_OK) e = hipErrorUnknown;
        if (e == hipSuccess) e = launch_gen_table_wide(c->d_gtab, wide_tab, small, bits, st0);
        if (e == hipSuccess) e = hipStreamSynchronize(st0);
        lap("two kernels + sync");
        if (e == hipSuccess) {
            g.tabs.push_back(GtabShared{c->device, bits, wide_tab, small, 1});
            c->d_gtab16 = wide_tab;
            c->gtab_bits = bits;
            break;
        }
        (void)hipGetLastError();
        if (wide_tab) (void)hipFree(wide_tab);
        if (small) (void)hipFree(small);
        if (why.empty()) why = std::string(hipGetErrorString(e)) + " at " + std::to_string(bits) + " bits";
    }
    if (!c->d_gtab16) c->gtab_wide_failed = true;
    if (!why.empty())
        c->gtab_note = "wide generator table unavailable (" + why + "): continuing on " +
                       (c->d_gtab16 ? "a " + std::to_string(c->gtab_bits) + "-bit table" : std::string("the 8-bit table"));
}

// ---- replacing the table in use by a wider one, without pausing anything --------------------------------------------------------
// (round 4 drained the frames, freed the old table — hipFree synchronises the device — and built the new one before the next
//  dispatch: 0.7 - 2.3 s of pause for 29 bits, and no way back when the 138 GB allocation then failed.)  Now: the two buffers are
// allocated by a thread of their own; the build then advances one SLICE per dispatch, launched in front of the dispatch's own
// kernels on the frame's own stream (a stream of its own would have to share a hardware queue with a frame and hold that
// frame's dispatches up behind a 0.2 s kernel: all twelve queues are taken); phase 1 starts when every slice of phase 0 has
// completed; the first dispatch after the last slice has completed takes the new table, and the old one is freed when the
// context next has nothing in flight.  A slice is sized to ~1/8 of the dispatch it rides with, so a scan runs ~12 % slower
// for the ~4 000 dispatches (3 s at 2^20 keys each) the 29-bit table takes, and never stops.

// Starts the background build of `bits` (the caller has checked that it is wanted, wider than the table in use and allowed).
void gtab_job_start(vgen_ctx *c, uint32_t bits) {
    GtabCache &g = gtab_cache();
    {
        std::lock_guard<std::mutex> lk(g.mu);
        if (std::find(g.building.begin(), g.building.end(), std::make_pair(c->device, bits)) != g.building.end()) {
            c->gtab_waiting_bits = bits;   // somebody else's build: take it when it is there
            return;
        }
        g.building.emplace_back(c->device, bits);
    }
    c->gtab_job.reset(new vgen_ctx::GtabJob());
    vgen_ctx::GtabJob *j = c->gtab_job.get();
    j->bits = bits;
    const int device = c->device;
    j->alloc = std::thread([j, device, bits]() {
        const uint32_t hook_fail_from = gtab_hook_fail_from();
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hook_fail_from && bits >= hook_fail_from ? hipErrorOutOfMemory : hipMalloc((void **)&j->wide, (size_t)ec_table_words(bits) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&j->small, (size_t)ec_table_small_words(bits) * sizeof(uint32_t));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            j->why = std::string(hipGetErrorString(e)) + " at " + std::to_string(bits) + " bits";
            j->alloc_state.store(2);
        } else {
            j->alloc_state.store(1);
        }
    });
}

// The widest table the context may move to from the one in use: `want`, or the next width towards the one in use that the memory
// policy allows (0 = none).  Both tables exist side by side until the old one can be freed.
uint32_t gtab_upgrade_target(vgen_ctx *c, uint32_t want, bool by_name, std::string *why) {
    std::vector<uint32_t> order{want};
    for (uint32_t b : GTAB_WIDTHS)
        if (b != want && gtab_additions(b) >= gtab_additions(want)) order.push_back(b);
    for (uint32_t bits : order) {
        if (gtab_additions(bits) >= gtab_additions(c->gtab_bits)) break;   // no better than what is in use
        if (bits == c->gtab_job_failed_bits) continue;
        std::string refused;
        if (mem_allows(c, gtab_bytes(bits), true, by_name && bits == want, &refused)) return bits;
        if (why && why->empty()) *why = refused + " at " + std::to_string(bits) + " bits";
    }
    return 0;
}

// One step of the background build, called on the dispatch path with the frame about to be dispatched to (its stream exists, it
// has nothing in flight).
int gtab_job_step(vgen_ctx *c, vgen_ctx::Frame &f) {
    if (c->gtab_waiting_bits) {   // another context's build of the width this one wants
        if (uint32_t *t = gtab_adopt_shared(c->device, c->gtab_waiting_bits)) {
            if (c->gtab_old) gtab_unref(c->device, t);   // (still holding a retired table: keep it simple, try again later)
            else {
                c->gtab_old = c->d_gtab16;
                c->d_gtab16 = t;
                c->gtab_bits = c->gtab_waiting_bits;
                c->gtab_waiting_bits = 0;
                c->gtab_note.clear();
            }
        } else {
            GtabCache &g = gtab_cache();
            std::lock_guard<std::mutex> lk(g.mu);
            if (std::find(g.building.begin(), g.building.end(), std::make_pair(c->device, c->gtab_waiting_bits)) == g.building.end())
                c->gtab_waiting_bits = 0;   // the builder gave up (or went away): decide afresh
        }
        return VGEN_OK;
    }
    if (!c->gtab_job) return VGEN_OK;
    vgen_ctx::GtabJob &j = *c->gtab_job;
    const int st = j.alloc_state.load();
    if (st == 0) return VGEN_OK;   // still allocating
    if (st == 2) {
        c->gtab_job_failed_bits = j.bits;
        c->gtab_note = "wider generator table unavailable (" + j.why + "): continuing on the " + std::to_string(c->gtab_bits) + "-bit table";
        gtab_job_drop(c);
        return VGEN_OK;
    }
    if (j.phase < 2) {
        const unsigned long long total = gen_table_phase_lanes(j.bits, j.phase);
        if (j.next < total) {
            // ~1/8 of the dispatch's own work: a lane of phase 0 is a whole 8-bit multiplication + inversion (~3 keys' worth of a
            // scalar-multiplication dispatch), a lane of phase 1 makes eight entries (~2 keys' worth)
            // ... but never less than one wave per SIMD of the device (phase 0: half a wave): a small dispatch is a chain of
            // latency-bound launches on a mostly idle device, where such a slice costs its own latency (~0.1 ms) and nothing else —
            // sized by the batch alone, a context of 8 192-key dispatches needed 300 000 of them for the 27-bit table
            unsigned long long count = j.phase == 0 ? std::max<unsigned long long>(c->batch / 32, 32768) : std::max<unsigned long long>(c->batch / 16, 65536);
            count = std::max<unsigned long long>(256, count & ~255ull);
            HIP_TRY(c, launch_gen_table_slice(c->d_gtab, j.wide, j.small, j.bits, j.phase, j.next, count, f.s));
            j.next += count;
            j.slices++;
            j.pending++;
            f.carries_slice = true;
        } else if (j.pending == 0) {
            j.phase++;
            j.next = 0;
        }
    }
    if (j.phase == 2 && !c->gtab_old) {
        // complete: publish it for the device's other contexts and take it into use from this dispatch on
        if (j.alloc.joinable()) j.alloc.join();
        {
            GtabCache &g = gtab_cache();
            std::lock_guard<std::mutex> lk(g.mu);
            g.tabs.push_back(GtabShared{c->device, j.bits, j.wide, j.small, 1});
            auto it = std::find(g.building.begin(), g.building.end(), std::make_pair(c->device, j.bits));
            if (it != g.building.end()) g.building.erase(it);
        }
        c->gtab_old = c->d_gtab16;
        c->d_gtab16 = j.wide;
        c->gtab_bits = j.bits;
        c->gtab_note.clear();
        if (c->trace_create) fprintf(stderr, "[generator tables] %u-bit table in use after %llu slices\n", j.bits, (unsigned long long)j.slices);
        c->gtab_job.reset();
    }
    return VGEN_OK;
}

// Generator tables of the paths that multiply a scalar per key.  `wide`: the dispatch is worth the wide-window table
// (every P2TR dispatch, arbitrary-scalar dispatches of a few thousand keys or more); a handful of keys, or the rare
// sequential batch that touches the group order, runs on the always-present 8-bit table instead of paying 11.8 GB and
// ~19 ms for it.  A wide table that cannot be had (allocation or build failure, the memory policy) is not an error either: the
// context steps down through the other widths in order of (additions per multiplication, bytes) before it settles on the 8-bit
// table (31), and says what it got through vgen_get_resources (never through vgen_last_error: the call succeeded).
// `f`: the frame being dispatched to (carries the background build's next slice).
int ensure_gtab(vgen_ctx *c, bool wide, vgen_ctx::Frame *f) {
    const bool trace = c->trace_create;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tlast = now();
    auto lap = [&](const char *what) { if (trace) { double t = now(); fprintf(stderr, "[generator tables] %-28s %.2f ms\n", what, t - tlast); tlast = t; } };
    if (!c->d_gtab) {
        std::vector<uint32_t> tab;
        host_gen_table8_limbs(tab);   // 8-bit windows, 652 800 B (core/ec.h)
        lap("host: 8-bit table");
        if (!mem_allows(c, tab.size() * sizeof(uint32_t), false, false, nullptr)) return c->fail(VGEN_E_NOMEM, "generator table: device_mem_budget_bytes exhausted");
        HIP_TRY(c, hipMalloc((void **)&c->d_gtab, tab.size() * sizeof(uint32_t)));
        if (int rc = upload(c, c->d_gtab, tab.data(), tab.size() * sizeof(uint32_t))) {
            (void)hipFree(c->d_gtab);
            c->d_gtab = nullptr;
            return rc;
        }
        c->mode_bytes += tab.size() * sizeof(uint32_t);
        lap("upload");
    }
    gtab_release_old_if_idle(c);
    if (!wide) return VGEN_OK;
    // The wide-window table, built on the device from the 8-bit one, once per DEVICE: 24-bit windows by default (11 windows,
    // 10 additions per multiplication instead of the 8-bit table's 31; 11.8 GB of the 288 GB).  Every entry is the affine sum
    // of two entries of a table of half the width, eight entries per lane sharing an inversion (gen_table_combine_kernel).
    // Measured in round 3 (profiles/r03_gtab26.txt; random-key mode Mkeys/s, build incl. allocation): 22 bits (3.2 GB) 1272,
    // 6 ms; 24 bits 1362, 19 ms; 26 bits (43 GB, 9 additions) 1451, but its first hipMalloc takes ~1 s and the gathers begin
    // to show (VALU-busy 0.97).  The paths are issue-bound (profiles/pmc_keys.json), so the rate follows the additions saved.
    bool by_name = false;
    const uint32_t want = wanted_table_bits(c, &by_name);
    if (!gtab_width_ok(want))
        return c->fail(VGEN_E_INVALID, "generator table width must be 8, 16, 20, 22, 24 or 26 (unsigned windows) or 25, 27 or 29 (signed windows)");
    if (!c->d_gtab16 && !c->gtab_wide_failed) {
        c->gtab_bits_wanted = want;
        if (want == 8) {
            c->gtab_wide_failed = true;   // (asked for: nothing to build)
            return VGEN_OK;
        }
        // A width somebody named is built at once (tests, bench legs: what they then measure is that width).  The scan loop's own
        // choice starts on the 24-bit table (there in ~20 ms) when it is the 29-bit one (0.7 - 2.3 s to allocate and build), which
        // then arrives in the background; 27 bits (~60 ms) is not worth two builds.
        uint32_t first = want;
        if (!by_name && gtab_additions(want) < gtab_additions(27) && !gtab_adopt_shared_peek(c->device, want)) first = 24;
        gtab_build_first(c, first, by_name, lap);
    }
    if (c->d_gtab16 && gtab_additions(want) < gtab_additions(c->gtab_bits)) {
        // a WIDER table is wanted than the one in use (a longer scan than the context has seen so far; a context never steps back
        // down by itself)
        c->gtab_bits_wanted = want;
        if (!c->gtab_job && !c->gtab_waiting_bits && !c->gtab_old) {
            std::string why;
            const uint32_t target = gtab_upgrade_target(c, want, by_name, &why);
            if (target) {
                if (uint32_t *t = gtab_adopt_shared(c->device, target)) {   // the device has it already
                    c->gtab_old = c->d_gtab16;
                    c->d_gtab16 = t;
                    c->gtab_bits = target;
                    c->gtab_note.clear();
                } else {
                    gtab_job_start(c, target);
                }
            }
            if (target != want && !why.empty() && c->gtab_note.empty())
                c->gtab_note = "wider generator table not taken (" + why + "): " + (target ? "moving to " + std::to_string(target) + " bits" : "staying on the " + std::to_string(c->gtab_bits) + "-bit table");
        }
    }
    if (f && (c->gtab_job || c->gtab_waiting_bits)) return gtab_job_step(c, *f);
    return VGEN_OK;
}

// The taproot stage of the arbitrary-scalar path on frame f, for n keys whose internal keys keys_bwd_kernel<P2TR> parked in the
// frame's `pts`: output / filter fields as the dispatch has them (`proto`), scratch from the frame's slice of the keys slab.
int enqueue_p2tr_stage(vgen_ctx *c, vgen_ctx::Frame &f, const KeysArgs &proto, uint32_t n) {
    KeysArgs t = proto;
    const size_t g = p2tr_groups(c);
    t.keys_be = nullptr;
    t.n = n;
    t.groups = (n + KEYS_WG - 1) / KEYS_WG;
    t.pts = f.d_keys_p2tr;
    t.xyz = t.pts + (size_t)c->batch * 16;
    t.tree = t.xyz + (size_t)27 * g * KEYS_WG;
    t.root = t.tree + g * 9 * KEYS_WG;
    // (the scratch is laid out for the full batch; a shorter dispatch uses its first t.groups workgroups' worth — the
    //  limb-major xz array is indexed with t.groups * 256 lanes by the kernels, which fits inside the full-batch region)
    HIP_TRY(c, launch_p2tr_tweak(t, f.s));
    return VGEN_OK;
}

// Keys and scratch of the arbitrary-scalar path for all frames (keys [batch][32 B] | xyz | tree | root per
// frame), one allocation when the path is first used: it serves vgen_dispatch_keys and the rare sequential
// batches that touch the group order, so most contexts never pay its ~155 MB per frame.
int ensure_keys_slab(vgen_ctx *c) {
    if (c->d_keys_slab) return VGEN_OK;
    const uint32_t max_groups = (c->batch + KEYS_WG - 1) / KEYS_WG;
    const size_t keys_b = up256((size_t)c->batch * 32);
    const size_t scratch_b = up256(((size_t)27 * max_groups * KEYS_WG + (size_t)max_groups * 9 * KEYS_WG + (size_t)9 * max_groups) * sizeof(uint32_t));
    const size_t stage_b = up256(p2tr_stage_words(c) * sizeof(uint32_t));   // taproot contexts: internal keys + the stage's own scratch
    const size_t per = keys_b + scratch_b + stage_b;
    std::string refused;
    if (!mem_allows(c, (uint64_t)per * c->frames, false, false, &refused)) return c->fail(VGEN_E_NOMEM, "arbitrary-scalar buffers: " + refused);
    HIP_TRY(c, hipMalloc((void **)&c->d_keys_slab, per * c->frames));
    c->mode_bytes += (uint64_t)per * c->frames;
    for (uint32_t i = 0; i < c->frames; i++) {
        c->fr[i].d_keys = c->d_keys_slab + per * i;
        c->fr[i].d_keys_scratch = reinterpret_cast<uint32_t *>(c->d_keys_slab + per * i + keys_b);
        c->fr[i].d_keys_p2tr = stage_b ? reinterpret_cast<uint32_t *>(c->d_keys_slab + per * i + keys_b + scratch_b) : nullptr;
    }
    return VGEN_OK;
}

// Enqueues the arbitrary-scalar kernels on frame f (its stream carries the whole chain): explicit keys
// (keys_dev != nullptr) or base + i.
int enqueue_keys(vgen_ctx *c, vgen_ctx::Frame &f, const uint8_t *keys_dev, const Scalar *base, uint32_t n) {
    // worth the wide table: taproot contexts (their sequential path builds it anyway) and real arbitrary-scalar batches
    // (a width named by the caller — vgen_params.table_bits, the test override — is honoured whatever the batch: the parity tests of
    //  every width rely on it)
    if (int rc = ensure_gtab(c, c->format == VGF_P2TR || (keys_dev != nullptr && n >= 4096) || c->gtab_env || c->gtab_param, &f)) return rc;
    if (int rc = ensure_keys_slab(c)) return rc;
    KeysArgs a;
    memset(&a, 0, sizeof a);
    a.gtab = c->d_gtab;
    a.gtab16 = c->d_gtab16;
    a.gtab_bits = c->gtab_bits;
    a.keys_be = keys_dev;
    if (base)
        for (int i = 0; i < 8; i++) a.base[i] = base->w[i];
    a.filter = c->d_filter;
    a.n = n;
    a.fmt = c->format;
    const uint32_t max_groups = (c->batch + KEYS_WG - 1) / KEYS_WG;
    a.groups = (n + KEYS_WG - 1) / KEYS_WG;
    a.xyz = f.d_keys_scratch;
    a.tree = a.xyz + (size_t)27 * max_groups * KEYS_WG;
    a.root = a.tree + (size_t)max_groups * 9 * KEYS_WG;
    const bool dump = dump_mode(c);
    if (c->h_filter.kind == DEVF_DFA && !dump) a.dfa_bytes = c->h_filter.dfa_bytes;
    // endomorphism contexts: six images per scalar multiplication here too (see rt_dispatch for the LDS rule)
    const bool parks_y = c->format == VGF_P2PKH_UNCOMPRESSED || c->format == VGF_ETHEREUM;
    const bool endo_now = c->endo && !(a.dfa_bytes && parks_y && a.dfa_bytes + 2u * 9u * KEYS_WG * 4u > 64u * 1024u);
    a.endo = endo_now ? 1u : 0u;
    a.vstride = c->batch;
    if (dump) {
        if (int rc = ensure_dump_frame(c, (uint32_t)(&f - c->fr.data()))) return rc;
        if (n < c->batch) HIP_TRY(c, hipMemsetAsync(f.d_dump, 0, (size_t)c->batch * (endo_now ? 6 : 1) * c->payload_words * sizeof(uint32_t), f.s));
        a.dump = f.d_dump;
    } else {
        a.mhdr = reinterpret_cast<DevMatchHeader *>(f.d_match);
        a.mrec = reinterpret_cast<DevMatch *>(f.d_match + sizeof(DevMatchHeader));
        a.match_base = f.match_base;
        a.match_cap = c->match_cap;
        if (c->h_filter.kind == DEVF_DFA) {
            a.dfa_blob = c->h_filter.dfa_blob;
            a.dfa_bytes = c->h_filter.dfa_bytes;
        }
    }
    if (c->format == VGF_P2TR) a.pts = f.d_keys_p2tr;   // keys_bwd_kernel<P2TR> parks the affine internal keys there
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_start, f.s));
    HIP_TRY(c, launch_keys_scan((int)c->format, a, f.s, c->timing ? f.ev_mid : nullptr));
    if (c->format == VGF_P2TR)
        if (int rc = enqueue_p2tr_stage(c, f, a, n)) return rc;
    return finish_dispatch(c, f, dump, endo_now ? (uint64_t)n * 6 : n, endo_now);
}

}  // namespace

int rt_dispatch(vgen_ctx *c, uint32_t frame, const uint8_t start_key_be[32]) {
    if (frame >= c->frames || !start_key_be) return c->fail(VGEN_E_INVALID, "bad frame index / key");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    if (c->injected_fault()) return c->fail(VGEN_E_HIP, "injected device failure (vgen_debug_fail_after)");
    Scalar k0;
    scalar_from_be(k0, start_key_be);
    if (!scalar_is_valid(k0)) return c->fail(VGEN_E_RANGE, "start key is not a valid secp256k1 scalar");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_frame(c, frame)) return rc;
    f.start = k0;
    // The batched affine additions have no exceptional cases as long as every scalar involved stays
    // below n (SURVEY.md §7 "hard parts"): k0 + N + S < n.  The (astronomically rare) batches that touch
    // the top of the scalar range go through the complete per-key kernel instead; keys >= n yield nothing
    // there (increment_key -> None, src/gpu.rs:963).
    if (scalar_distance_to_n(k0) <= (uint64_t)c->batch + c->S) return enqueue_keys(c, f, nullptr, &k0, c->batch);

    // Q_j = (k0 + N/2 - S/2 + j) * G, j < S, passed by value in the kernel arguments
    const uint32_t S = c->S;
    Scalar kb;
    scalar_add_u64(kb, k0, (uint64_t)c->batch / 2 - S / 2);
    ge aff[32];
    if (!host_seq_points(c->base_cache, kb, S, aff)) return c->fail(VGEN_E_RANGE, "base point at infinity");

    SeqArgs a;
    memset(&a, 0, sizeof a);
    for (uint32_t j = 0; j < S; j++) {
        DevSeqQ &q = a.q[j];
        fe nx, ny;
        fe_canon_neg(nx, aff[j].x);
        fe_canon_neg(ny, aff[j].y);
        for (int i = 0; i < 9; i++) {
            q.qx[i] = aff[j].x.n[i];
            q.qy[i] = aff[j].y.n[i];
            q.nqx[i] = nx.n[i];
            q.nqy[i] = ny.n[i];
        }
    }
    a.rtab = c->d_rtab;
    a.filter = c->d_filter;
    a.pre = f.d_scratch;
    a.tree = a.pre + (size_t)S * 9 * c->lanes;
    a.root = a.tree + (size_t)c->groups * 9 * SEQ_WG;
    if (split_format(c)) {
        a.xs = a.root + (size_t)9 * c->groups;
        a.hash_kpl = c->hash_kpl;
    }
    a.lanes = c->lanes;
    a.groups = c->groups;
    a.n = c->batch;
    a.s = S;
    // The twin without priority changes (hipcc's schedule of the hash pair) serves launches that will have their SIMDs (nearly) to themselves: decided by what is
    // IN FLIGHT when this dispatch is issued — at most lone_max_others other frames of this context —, not by how many frames the
    // context was created with: a caller that keeps one or two dispatches going on a 12-frame context (the reference's own loop keeps
    // 2, src/gpu.rs:399) gets the kernel that is 17 % faster alone, and so do the first dispatches of a burst.
    if (c->lone_variant) {
        uint32_t others = 0;
        for (const vgen_ctx::Frame &g : c->fr) others += (&g != &f && g.in_flight) ? 1u : 0u;
        a.lone = others <= c->lone_max_others;
    }
    const bool dump = dump_mode(c);
    if (dump) {
        if (int rc = ensure_dump_frame(c, (uint32_t)(&f - c->fr.data()))) return rc;
        a.dump = f.d_dump;
    } else {
        a.mhdr = reinterpret_cast<DevMatchHeader *>(f.d_match);
        a.mrec = reinterpret_cast<DevMatch *>(f.d_match + sizeof(DevMatchHeader));
        a.match_base = f.match_base;   // the counter is monotonic: no reset, no upload
        a.match_cap = c->match_cap;
        if (c->h_filter.kind == DEVF_DFA) {
            a.dfa_blob = c->h_filter.dfa_blob;
            a.dfa_bytes = c->h_filter.dfa_bytes;
        }
    }
    a.fmt = c->format;
    // six images per point (every format but P2TR: c->endo is never set there), whatever the filter — unless the
    // pattern's automaton leaves no room in a workgroup's 64 KiB of LDS for the y coordinate the uncompressed / Ethereum
    // formats park beside the product tree (2 x 9 KiB static + the blob): such a dispatch tests the plain keys
    const bool parks_y = c->format == VGF_P2PKH_UNCOMPRESSED || c->format == VGF_ETHEREUM;
    const bool endo_now = c->endo && !(a.dfa_bytes && parks_y && a.dfa_bytes + 2u * 9u * SEQ_WG * 4u > 64u * 1024u);
    a.endo = endo_now ? 1u : 0u;
    if (c->format == VGF_P2TR) {   // the tweak multiplication t*G needs the fixed-window table ...
        if (int rc = ensure_gtab(c, true, &f)) return rc;
        a.gtab = c->d_gtab;
        a.gtab16 = c->d_gtab16;
        a.gtab_bits = c->gtab_bits;
        // ... and the tweaked points of the dispatch wait in scratch for their shared inversion:
        // tq [2S][27][lanes] | tq_flag [2S][lanes] | tree2 [groups][9][WG] | root2 [9][groups]
        const size_t tq_words = (size_t)2 * S * 27 * c->lanes, flag_words = (size_t)2 * S * c->lanes;
        const size_t tree_words = (size_t)c->groups * 9 * SEQ_WG;
        a.tq = f.d_p2tr_scratch;
        a.tq_flag = a.tq + tq_words;
        a.tree2 = a.tq_flag + flag_words;
        a.root2 = a.tree2 + tree_words;
    }
    // the whole chain (seq_fwd -> seq_inv -> seq_bwd [-> P2TR stages] -> result copy) on the frame's own stream
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_start, f.s));
    HIP_TRY(c, launch_seq_fwd(a, f.s));
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_mid, f.s));
    HIP_TRY(c, launch_seq_bwd((int)c->format, a, f.s));
    return finish_dispatch(c, f, dump, endo_now ? (uint64_t)c->batch * 6 : c->batch, endo_now);
}

int rt_dispatch_keys(vgen_ctx *c, uint32_t frame, const uint8_t *keys_be, uint32_t n) {
    if (frame >= c->frames || !keys_be) return c->fail(VGEN_E_INVALID, "bad frame index / key buffer");
    if (n == 0 || n > c->batch) return c->fail(VGEN_E_INVALID, "vgen_dispatch_keys: n must be in [1, batch_size]");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    if (c->injected_fault()) return c->fail(VGEN_E_HIP, "injected device failure (vgen_debug_fail_after)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_frame(c, frame)) return rc;
    if (int rc = ensure_keys_slab(c)) return rc;
    HIP_TRY(c, hipMemcpyAsync(f.d_keys, keys_be, (size_t)n * 32, hipMemcpyHostToDevice, f.s));
    memset(&f.start, 0, sizeof f.start);
    return enqueue_keys(c, f, f.d_keys, nullptr, n);
}

// Independent random keys (the reference CPU path's shape, src/scanner.rs:144-155) with no upload: lane i draws
// key(seed, stream, first_index + i) from the counter-based stream of core/rnd.h on the device.
int rt_dispatch_random(vgen_ctx *c, uint32_t frame, const RndSeed &seed, uint32_t stream, uint64_t first_index) {
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    if (first_index > UINT64_MAX - (c->batch - 1)) return c->fail(VGEN_E_RANGE, "vgen_dispatch_random: index range wraps 2^64");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    if (c->injected_fault()) return c->fail(VGEN_E_HIP, "injected device failure (vgen_debug_fail_after)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_frame(c, frame)) return rc;
    if (int rc = ensure_keys_slab(c)) return rc;
    memset(&f.start, 0, sizeof f.start);
    // the scalars are drawn into the frame's key buffer by a small kernel; from there on the dispatch is vgen_dispatch_keys'
    HIP_TRY(c, launch_rnd_fill(f.d_keys, c->batch, seed, stream, first_index, f.s));
    return enqueue_keys(c, f, f.d_keys, nullptr, c->batch);
}

int rt_wait(vgen_ctx *c, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches, uint64_t *keys_tested) {
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    vgen_ctx::Frame &f = c->fr[frame];
    if (!f.in_flight) return c->fail(VGEN_E_STATE, "No pending operation on frame " + std::to_string(frame));
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = wait_done(c, f)) return rc;
    f.in_flight = false;
    if (f.carries_slice) {   // the slice of the background table build that rode in front of this dispatch has completed with it
        f.carries_slice = false;
        if (c->gtab_job && c->gtab_job->pending) c->gtab_job->pending--;
    }
    f.timing_fresh = false;   // elapsed times are read from the events on demand (rt_frame_times)
    if (keys_tested) *keys_tested = f.keys_tested;
    uint32_t found = 0;
    if (!f.dumped) {
        const DevMatchHeader *hdr = reinterpret_cast<const DevMatchHeader *>(f.h_match);
        found = hdr->count - f.match_base;   // mod 2^32
        f.match_base = hdr->count;
        f.last_clk_cycles = hdr->clk_cycles - f.clk_cycles_seen;
        f.last_clk_ticks = hdr->clk_ticks - f.clk_ticks_seen;
        f.clk_cycles_seen = hdr->clk_cycles;
        f.clk_ticks_seen = hdr->clk_ticks;
        uint32_t stored = std::min(found, c->match_cap);
        if (stored > FIRST_COPY) {   // rare: the tail of a busy ring, fetched on the frame's own stream
            HIP_TRY(c, hipMemcpyAsync(f.h_match + match_bytes(FIRST_COPY), f.d_match + match_bytes(FIRST_COPY),
                                      (size_t)(stored - FIRST_COPY) * sizeof(DevMatch), hipMemcpyDeviceToHost, f.s));
            if (!frame_owns_stream(c)) HIP_TRY(c, hipEventRecord(f.ev_done, f.s));
            if (int rc = wait_done(c, f)) return rc;
        }
        const DevMatch *rec = reinterpret_cast<const DevMatch *>(f.h_match + sizeof(DevMatchHeader));
        // ascending index, the order the reference's par_iter().enumerate() collect yields (gpu.rs:1030-1093).  The device hands
        // out slots in arrival order; what is sorted is one 64-bit word per record (index, slot) — a permissive pattern brings
        // tens of thousands of 40-byte records per dispatch, and sorting those in place was most of this call — and the
        // records are then copied out once, in order.
        if (out) {
            std::vector<uint64_t> &order = c->sort_scratch;
            order.resize(stored);
            for (uint32_t i = 0; i < stored; i++) order[i] = (uint64_t)rec[i].index << 32 | i;
            if (stored < 2048) {
                std::sort(order.begin(), order.end());
            } else {
                // three stable counting passes over 11 bits of the index each (a comparison sort of ~45 000 words per dispatch was
                // 2 ms of every `^1C` batch)
                std::vector<uint64_t> &tmp = c->sort_scratch2;
                tmp.resize(stored);
                uint64_t *src = order.data(), *dst = tmp.data();
                for (int pass = 0; pass < 3; pass++) {
                    const int sh = 32 + 11 * pass;
                    uint32_t hist[2049] = {0};
                    for (uint32_t i = 0; i < stored; i++) hist[((src[i] >> sh) & 2047u) + 1]++;
                    for (int b = 0; b < 2048; b++) hist[b + 1] += hist[b];
                    for (uint32_t i = 0; i < stored; i++) dst[hist[(src[i] >> sh) & 2047u]++] = src[i];
                    std::swap(src, dst);
                }
                if (src != order.data()) memcpy(order.data(), src, (size_t)stored * sizeof(uint64_t));   // (three passes: the result is in tmp)
            }
            const uint32_t n = std::min(stored, cap);
            for (uint32_t i = 0; i < n; i++) {
                const DevMatch &r = rec[(uint32_t)order[i]];
                out[i].index = r.index;
                out[i].reserved = 0;
                memcpy(out[i].payload, r.payload, 32);
            }
        }
    }
    if (n_matches) *n_matches = found;
    return VGEN_OK;
}

// Durations of the frame's last completed dispatch: the dominant kernel (seq_bwd / keys) and the whole dispatch.
int rt_frame_times(vgen_ctx *c, uint32_t frame, float *kernel_ms, float *total_ms) {
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame still in flight");
    if (!c->timing) return c->fail(VGEN_E_STATE, "the context was created without VGEN_FLAG_TIMING");
    if (!f.timing_fresh && f.s) {
        HIP_TRY(c, hipSetDevice(c->device));
        (void)hipEventElapsedTime(&f.last_ms, f.ev_mid, f.ev_stop);
        (void)hipEventElapsedTime(&f.last_total_ms, f.ev_start, f.ev_stop);
        f.timing_fresh = true;
    }
    if (kernel_ms) *kernel_ms = f.last_ms;
    if (total_ms) *total_ms = f.last_total_ms;
    return VGEN_OK;
}

// Shader-clock sample of the frame's last completed filter-mode dispatch (kernels.hip): cycles / ticks * 100 = MHz.
int rt_frame_clock(vgen_ctx *c, uint32_t frame, uint32_t *cycles, uint32_t *ticks) {
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame still in flight");
    if (cycles) *cycles = f.dumped ? 0 : f.last_clk_cycles;
    if (ticks) *ticks = f.dumped ? 0 : f.last_clk_ticks;
    return VGEN_OK;
}

// The pinned mirror of the frame's dump (filled by the dispatch's own copy): valid until the frame is dispatched again.
int rt_dump_view(vgen_ctx *c, uint32_t frame, const uint8_t **ptr, size_t *len) {
    if (frame >= c->frames || !ptr) return c->fail(VGEN_E_INVALID, "bad frame index / pointer");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "vgen_read_dump before vgen_wait");
    if (!f.dumped || !f.h_dump) return c->fail(VGEN_E_STATE, "frame's last dispatch was not in dump mode");
    *ptr = f.h_dump;
    if (len) *len = (size_t)f.dump_slots * c->payload_words * sizeof(uint32_t);
    return VGEN_OK;
}

int rt_read_dump(vgen_ctx *c, uint32_t frame, uint8_t *out, size_t out_len) {
    if (!out) return c->fail(VGEN_E_INVALID, "bad frame index / buffer");
    const uint8_t *src = nullptr;
    size_t need = 0;
    if (int rc = rt_dump_view(c, frame, &src, &need)) return rc;
    if (out_len < need) return c->fail(VGEN_E_INVALID, "output buffer too small");
    memcpy(out, src, need);
    return VGEN_OK;
}

void rt_prefer_table_bits(vgen_ctx *c, uint32_t bits, uint32_t cap) {
    if (bits == 0 || (bits != 8 && gtab_width_ok(bits))) c->gtab_bits_pref = bits;
    c->gtab_bits_cap = cap && gtab_width_ok(cap) ? cap : 0;
}

// Device and pinned host memory of the context, and what the device has left: vgen_get_memory.
int rt_get_memory(const vgen_ctx *c, vgen_memory_info *out) {
    if (!out || out->struct_size != sizeof(vgen_memory_info)) return VGEN_E_INVALID;
    out->table_bits = c->d_gtab16 ? c->gtab_bits : c->d_gtab ? 8u : 0u;
    out->frames_bytes = c->frames_bytes;
    out->mode_bytes = c->mode_bytes;
    out->table_bytes = gtab_held_bytes(c);
    out->pinned_host_bytes = c->pinned_bytes;
    out->budget_bytes = c->mem_budget;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
    out->device_free_bytes = free_b;
    out->device_total_bytes = total_b;
    return VGEN_OK;
}

// What dump mode and the scalar-multiplication paths have (or will get) of what they ask for: vgen_get_resources.
int rt_get_resources(const vgen_ctx *c, uint32_t *dump_frames, uint32_t *table_bits, uint32_t *table_bits_wanted, std::string *note) {
    if (dump_frames) *dump_frames = dump_frames_for(c);
    // 0 = no generator table yet (no P2TR / arbitrary-scalar dispatch so far); 8 = the 8-bit table only
    if (table_bits) *table_bits = c->d_gtab16 ? c->gtab_bits : c->d_gtab ? 8u : 0u;
    if (table_bits_wanted) *table_bits_wanted = c->gtab_bits_wanted ? c->gtab_bits_wanted : wanted_table_bits(c, nullptr);
    if (note) *note = c->gtab_note;
    return VGEN_OK;
}

}  // namespace vg
