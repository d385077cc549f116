// runtime.cpp — see runtime.h.
#include "runtime.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

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

// pre [S][9][lanes] | tree [groups][9][WG] | root [9][groups]
size_t scratch_words(const vgen_ctx *c) {
    return (size_t)c->S * 9 * c->lanes + (size_t)c->groups * 9 * SEQ_WG + (size_t)9 * c->groups;
}

void fe_canon_neg(fe &r, const fe &a) {
    fe_neg(r, a, 1);
    fe_normalize(r);
}

}  // namespace

int rt_create(const vgen_params *p, vgen_ctx **out, std::string &err) {
    if (!p || !out || p->struct_size != sizeof(vgen_params)) {
        err = "vgen_create: bad parameter block";
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
    vgen_ctx *c = new vgen_ctx();
    c->device = p->device;
    c->batch = p->batch_size ? p->batch_size : (1u << 20);
    c->frames = p->frames ? p->frames : 2;
    c->match_cap = p->match_cap ? p->match_cap : 4096;
    c->format = p->format;
    c->payload_words = p->format == VGF_P2TR ? 8 : 5;
    c->timing = (p->flags & VGEN_FLAG_TIMING) != 0;
    c->S = env_u32("VGEN_SEQ_S", 8);
    auto bail = [&](int st, const std::string &m) {
        err = m;
        rt_destroy(c);
        return st;
    };
    if (c->frames > 20) return bail(VGEN_E_INVALID, "frames must be <= 20 (more streams than hardware queues collapse the throughput)");
    if (c->S < 2 || c->S > SEQ_MAX_S || (c->S & (c->S - 1))) return bail(VGEN_E_INVALID, "VGEN_SEQ_S must be a power of two in [2, 16]");
    if (c->batch % 8192 != 0 || c->batch % (2 * SEQ_WG * c->S) != 0 || c->batch < 8192)
        return bail(VGEN_E_INVALID, "batch_size must be a multiple of 8192 (and of 512*S)");
    if (c->match_cap < FIRST_COPY) c->match_cap = FIRST_COPY;
    c->lanes = c->batch / (2 * c->S);
    c->groups = c->lanes / SEQ_WG;

    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return bail(VGEN_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));

    // offset table R_u = (u*S + S/2) * G, uploaded limb-major ([18][lanes]) for coalesced reads
    {
        std::vector<ge> tab;
        host_build_stride_table(c->S / 2, c->S, c->lanes, tab);
        std::vector<uint32_t> lm((size_t)18 * c->lanes);
        for (uint32_t u = 0; u < c->lanes; u++)
            for (int i = 0; i < 9; i++) {
                lm[(size_t)i * c->lanes + u] = tab[u].x.n[i];
                lm[(size_t)(9 + i) * c->lanes + u] = tab[u].y.n[i];
            }
        e = hipMalloc((void **)&c->d_rtab, lm.size() * sizeof(uint32_t));
        if (e != hipSuccess) return bail(VGEN_E_NOMEM, std::string("hipMalloc(rtab): ") + hipGetErrorString(e));
        e = hipMemcpy(c->d_rtab, lm.data(), lm.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) return bail(VGEN_E_HIP, std::string("hipMemcpy(rtab): ") + hipGetErrorString(e));
    }
    e = hipMalloc((void **)&c->d_filter, sizeof(DevFilter));
    if (e != hipSuccess) return bail(VGEN_E_NOMEM, std::string("hipMalloc(filter): ") + hipGetErrorString(e));

    // One device slab and one pinned slab for all frames: sixteen frames as separate allocations cost ~85 ms of
    // vgen_create (time to first match, cold); slices are 256-byte aligned.
    c->fr.resize(c->frames);
    auto up256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t scratch_b = up256(scratch_words(c) * sizeof(uint32_t)), match_b = up256(match_bytes(c->match_cap));
    if ((e = hipMalloc((void **)&c->d_slab, (scratch_b + match_b) * c->frames)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_slab, match_b * c->frames, hipHostMallocDefault)) != hipSuccess)
        return bail(VGEN_E_NOMEM, std::string("frame allocation: ") + hipGetErrorString(e));
    for (uint32_t i = 0; i < c->frames; i++) {
        vgen_ctx::Frame &f = c->fr[i];
        f.d_scratch = reinterpret_cast<uint32_t *>(c->d_slab + (scratch_b + match_b) * i);
        f.d_match = c->d_slab + (scratch_b + match_b) * i + scratch_b;
        f.h_match = c->h_slab + match_b * i;
        // (the frame's stream and events are created on its first dispatch: a stream costs ~5 ms, and a scan's
        // first frames should be running while the later ones are still being set up)
    }
    // The match rings start at zero (monotonic counters); the scratch needs no initialisation.  The frames'
    // streams do not synchronise with the null stream, so the clears must have finished before vgen_create returns.
    for (uint32_t i = 0; i < c->frames; i++)
        if ((e = hipMemset(c->fr[i].d_match, 0, match_bytes(c->match_cap))) != hipSuccess)
            return bail(VGEN_E_NOMEM, std::string("frame setup: ") + hipGetErrorString(e));
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(VGEN_E_HIP, std::string("frame setup: ") + hipGetErrorString(e));
    *out = c;
    return VGEN_OK;
}

void rt_destroy(vgen_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &f : c->fr) {
        if (f.stream) (void)hipStreamSynchronize(f.stream);
        if (f.d_dump) (void)hipFree(f.d_dump);
        if (f.d_keys) (void)hipFree(f.d_keys);
        if (f.d_keys_scratch) (void)hipFree(f.d_keys_scratch);
        if (f.d_p2tr_scratch) (void)hipFree(f.d_p2tr_scratch);
        if (f.ev_start) (void)hipEventDestroy(f.ev_start);
        if (f.ev_mid) (void)hipEventDestroy(f.ev_mid);
        if (f.ev_stop) (void)hipEventDestroy(f.ev_stop);
        if (f.stream) (void)hipStreamDestroy(f.stream);
    }
    if (c->probe_stream) {
        (void)hipStreamSynchronize(c->probe_stream);
        (void)hipStreamDestroy(c->probe_stream);
    }
    if (c->d_probe) (void)hipFree(c->d_probe);
    if (c->d_slab) (void)hipFree(c->d_slab);      // scratch + match rings of all frames
    if (c->h_slab) (void)hipHostFree(c->h_slab);
    if (c->d_rtab) (void)hipFree(c->d_rtab);
    if (c->d_gtab) (void)hipFree(c->d_gtab);
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
    HIP_TRY(c, hipStreamSynchronize(c->probe_stream));
    c->probe_running = false;
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(v, c->d_probe, sizeof v, hipMemcpyDeviceToHost));
    if (v[1] == 0) return c->fail(VGEN_E_HIP, "clock probe returned no ticks");
    *mhz = (double)v[0] / (double)v[1] * 100.0;
    return VGEN_OK;
}

int rt_set_filter(vgen_ctx *c, const vgen_filter *f) {
    HIP_TRY(c, hipSetDevice(c->device));
    for (auto &fr : c->fr)
        if (fr.in_flight) return c->fail(VGEN_E_STATE, "vgen_set_filter while a dispatch is in flight");
    if (!f) {
        c->have_filter = false;
        return VGEN_OK;
    }
    if (f->format != c->format) return c->fail(VGEN_E_INVALID, "filter was compiled for another address format");
    c->h_filter = f->dev;
    if (f->dev.chk_lut) {   // Bech32 checksum tables: upload and point the device copy at them
        if (!c->d_chk_lut) HIP_TRY(c, hipMalloc((void **)&c->d_chk_lut, 32 * 256 * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(c->d_chk_lut, f->chk_lut.data(), f->chk_lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->h_filter.chk_lut = c->d_chk_lut;
    }
    if (f->dev.kind == DEVF_DFA) {   // the pattern's automaton: upload, point the device copy at it
        if (!c->d_dfa) HIP_TRY(c, hipMalloc((void **)&c->d_dfa, 48 * 1024));
        HIP_TRY(c, hipMemcpy(c->d_dfa, f->dfa_blob.data(), f->dfa_blob.size() * 4, hipMemcpyHostToDevice));
        c->h_filter.dfa_blob = c->d_dfa;
    }
    HIP_TRY(c, hipMemcpy(c->d_filter, &c->h_filter, sizeof(DevFilter), hipMemcpyHostToDevice));
    // the uploads ran on the null stream, which the frames' non-blocking streams do not wait for
    HIP_TRY(c, hipDeviceSynchronize());
    c->have_filter = true;
    return VGEN_OK;
}

namespace {

// The frame's stream and timing events, created on first use.
int ensure_stream(vgen_ctx *c, vgen_ctx::Frame &f) {
    if (f.stream) return VGEN_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreate(&f.ev_start));
    HIP_TRY(c, hipEventCreate(&f.ev_mid));
    HIP_TRY(c, hipEventCreate(&f.ev_stop));
    return VGEN_OK;
}

// Enqueues the arbitrary-scalar kernel on frame f: explicit keys (keys_dev != nullptr) or base + i.
int ensure_gtab(vgen_ctx *c) {
    if (!c->d_gtab) {
        std::vector<uint32_t> tab;
        host_gen_table8_limbs(tab);   // 8-bit windows, 652 800 B (core/ec.h)
        HIP_TRY(c, hipMalloc((void **)&c->d_gtab, tab.size() * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(c->d_gtab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipDeviceSynchronize());   // (null-stream upload: see rt_set_filter)
    }
    return VGEN_OK;
}

int enqueue_keys(vgen_ctx *c, vgen_ctx::Frame &f, const uint8_t *keys_dev, const Scalar *base, uint32_t n) {
    if (int rc = ensure_gtab(c)) return rc;
    if (int rc = ensure_stream(c, f)) return rc;
    KeysArgs a;
    memset(&a, 0, sizeof a);
    a.gtab = c->d_gtab;
    a.keys_be = keys_dev;
    if (base)
        for (int i = 0; i < 8; i++) a.base[i] = base->w[i];
    a.filter = c->d_filter;
    a.n = n;
    a.fmt = c->format;
    // scratch of the three stages, sized for a full batch on first use: xyz | tree | root
    const uint32_t max_groups = (c->batch + KEYS_WG - 1) / KEYS_WG;
    if (!f.d_keys_scratch)
        HIP_TRY(c, hipMalloc((void **)&f.d_keys_scratch,
                             ((size_t)27 * max_groups * KEYS_WG + (size_t)max_groups * 9 * KEYS_WG + (size_t)9 * max_groups) * sizeof(uint32_t)));
    a.groups = (n + KEYS_WG - 1) / KEYS_WG;
    a.xyz = f.d_keys_scratch;
    a.tree = a.xyz + (size_t)27 * max_groups * KEYS_WG;
    a.root = a.tree + (size_t)max_groups * 9 * KEYS_WG;
    const bool dump = !c->have_filter || c->h_filter.kind == DEVF_HOST_ALL;
    if (dump) {
        if (!f.d_dump) HIP_TRY(c, hipMalloc((void **)&f.d_dump, (size_t)c->batch * c->payload_words * sizeof(uint32_t)));
        if (n < c->batch) HIP_TRY(c, hipMemsetAsync(f.d_dump, 0, (size_t)c->batch * c->payload_words * sizeof(uint32_t), f.stream));
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
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_start, f.stream));
    HIP_TRY(c, launch_keys_scan((int)c->format, a, f.stream, c->timing ? f.ev_mid : nullptr));
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_stop, f.stream));
    if (!dump)
        HIP_TRY(c, hipMemcpyAsync(f.h_match, f.d_match, match_bytes(FIRST_COPY), hipMemcpyDeviceToHost, f.stream));
    f.in_flight = true;
    f.dumped = dump;
    f.keys_tested = n;
    return VGEN_OK;
}

}  // namespace

int rt_dispatch(vgen_ctx *c, uint32_t frame, const uint8_t start_key_be[32]) {
    if (frame >= c->frames || !start_key_be) return c->fail(VGEN_E_INVALID, "bad frame index / key");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    Scalar k0;
    scalar_from_be(k0, start_key_be);
    if (!scalar_is_valid(k0)) return c->fail(VGEN_E_RANGE, "start key is not a valid secp256k1 scalar");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_stream(c, f)) return rc;
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
    a.lanes = c->lanes;
    a.groups = c->groups;
    a.n = c->batch;
    a.s = S;
    const bool dump = !c->have_filter || c->h_filter.kind == DEVF_HOST_ALL;
    if (dump) {
        if (!f.d_dump) HIP_TRY(c, hipMalloc((void **)&f.d_dump, (size_t)c->batch * c->payload_words * sizeof(uint32_t)));
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
    if (c->format == VGF_P2TR) {   // the tweak multiplication t*G needs the fixed-window table ...
        if (int rc = ensure_gtab(c)) return rc;
        a.gtab = c->d_gtab;
        // ... and the tweaked points of the dispatch wait in scratch for their shared inversion:
        // tq [2S][27][lanes] | tq_flag [2S][lanes] | tree2 [groups][9][WG] | root2 [9][groups]
        const size_t tq_words = (size_t)2 * S * 27 * c->lanes, flag_words = (size_t)2 * S * c->lanes;
        const size_t tree_words = (size_t)c->groups * 9 * SEQ_WG, root_words = (size_t)9 * c->groups;
        if (!f.d_p2tr_scratch)
            HIP_TRY(c, hipMalloc((void **)&f.d_p2tr_scratch, (tq_words + flag_words + tree_words + root_words) * sizeof(uint32_t)));
        a.tq = f.d_p2tr_scratch;
        a.tq_flag = a.tq + tq_words;
        a.tree2 = a.tq_flag + flag_words;
        a.root2 = a.tree2 + tree_words;
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_start, f.stream));
    HIP_TRY(c, launch_seq_scan((int)c->format, a, f.stream, c->timing ? f.ev_mid : nullptr));
    if (c->timing) HIP_TRY(c, hipEventRecord(f.ev_stop, f.stream));
    if (!dump)
        HIP_TRY(c, hipMemcpyAsync(f.h_match, f.d_match, match_bytes(FIRST_COPY), hipMemcpyDeviceToHost, f.stream));
    f.in_flight = true;
    f.dumped = dump;
    f.keys_tested = c->batch;
    return VGEN_OK;
}

int rt_dispatch_keys(vgen_ctx *c, uint32_t frame, const uint8_t *keys_be, uint32_t n) {
    if (frame >= c->frames || !keys_be) return c->fail(VGEN_E_INVALID, "bad frame index / key buffer");
    if (n == 0 || n > c->batch) return c->fail(VGEN_E_INVALID, "vgen_dispatch_keys: n must be in [1, batch_size]");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_stream(c, f)) return rc;
    if (!f.d_keys) HIP_TRY(c, hipMalloc((void **)&f.d_keys, (size_t)c->batch * 32));
    HIP_TRY(c, hipMemcpyAsync(f.d_keys, keys_be, (size_t)n * 32, hipMemcpyHostToDevice, f.stream));
    memset(&f.start, 0, sizeof f.start);
    return enqueue_keys(c, f, f.d_keys, nullptr, n);
}

int rt_wait(vgen_ctx *c, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches, uint64_t *keys_tested) {
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    vgen_ctx::Frame &f = c->fr[frame];
    if (!f.in_flight) return c->fail(VGEN_E_STATE, "No pending operation on frame " + std::to_string(frame));
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(f.stream));
    f.in_flight = false;
    f.timing_fresh = false;   // elapsed times are read from the events on demand (rt_frame_times)
    if (keys_tested) *keys_tested = f.keys_tested;
    uint32_t found = 0;
    if (!f.dumped) {
        const DevMatchHeader *hdr = reinterpret_cast<const DevMatchHeader *>(f.h_match);
        found = hdr->count - f.match_base;   // mod 2^32
        f.match_base = hdr->count;
        uint32_t stored = std::min(found, c->match_cap);
        if (stored > FIRST_COPY)
            HIP_TRY(c, hipMemcpy(f.h_match + match_bytes(FIRST_COPY), f.d_match + match_bytes(FIRST_COPY),
                                 (size_t)(stored - FIRST_COPY) * sizeof(DevMatch), hipMemcpyDeviceToHost));
        DevMatch *rec = reinterpret_cast<DevMatch *>(f.h_match + sizeof(DevMatchHeader));
        // ascending index, the order the reference's par_iter().enumerate() collect yields (gpu.rs:1030-1093)
        std::sort(rec, rec + stored, [](const DevMatch &x, const DevMatch &y) { return x.index < y.index; });
        if (out) {
            uint32_t n = std::min(stored, cap);
            for (uint32_t i = 0; i < n; i++) {
                out[i].index = rec[i].index;
                out[i].reserved = 0;
                memcpy(out[i].payload, rec[i].payload, 32);
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
    if (!f.timing_fresh && f.stream) {
        HIP_TRY(c, hipSetDevice(c->device));
        (void)hipEventElapsedTime(&f.last_ms, f.ev_mid, f.ev_stop);
        (void)hipEventElapsedTime(&f.last_total_ms, f.ev_start, f.ev_stop);
        f.timing_fresh = true;
    }
    if (kernel_ms) *kernel_ms = f.last_ms;
    if (total_ms) *total_ms = f.last_total_ms;
    return VGEN_OK;
}

int rt_read_dump(vgen_ctx *c, uint32_t frame, uint8_t *out, size_t out_len) {
    if (frame >= c->frames || !out) return c->fail(VGEN_E_INVALID, "bad frame index / buffer");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "vgen_read_dump before vgen_wait");
    if (!f.dumped || !f.d_dump) return c->fail(VGEN_E_STATE, "frame's last dispatch was not in dump mode");
    const size_t need = (size_t)c->batch * c->payload_words * sizeof(uint32_t);
    if (out_len < need) return c->fail(VGEN_E_INVALID, "output buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(out, f.d_dump, need, hipMemcpyDeviceToHost));
    return VGEN_OK;
}

}  // namespace vg
