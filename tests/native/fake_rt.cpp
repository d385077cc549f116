// fake_rt.cpp — TEST-ONLY stand-in for vgen_amd/csrc/runtime.cpp: the rt_* interface of runtime.h implemented on the
// CPU, so that the host side of libvgen_hip.so that needs a device to run — scanner.cpp (worker pool, helper-thread ramp,
// shared atomics, checkpoint lock, multi-context threads, failure take-over) and cabi.cpp — can be built with
// -fsanitize=thread and -fsanitize=address,undefined and exercised in the build container (no sanitizer runs on the
// GPU box).  SURVEY.md 5: "host ASan/TSan builds of the C++ host"; the reference gets the same guarantees from
// `unsafe_code = "forbid"` (Cargo.toml:14).
//
// What stands in for the kernels is the product's OWN single-source core (core/*.h as g++ compiles it: fe / ec / hash /
// filter_eval / dfa_eval / rnd) plus host/host_ec.cpp — never the oracle, which stays the checker (fake_driver.cpp
// compares every scan against it).  A dispatch is computed by a thread of its own and joined by rt_wait, so the frames
// really are concurrent with the scan loop, as on the device.  Not part of the product; never loaded by vgen_amd.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "../../vgen_amd/csrc/core/dfa_eval.h"
#include "../../vgen_amd/csrc/core/filter_eval.h"
#include "../../vgen_amd/csrc/core/hash.h"
#include "../../vgen_amd/csrc/core/rnd.h"
#include "../../vgen_amd/csrc/core/taproot.h"
#include "../../vgen_amd/csrc/host/encode.h"
#include "../../vgen_amd/csrc/runtime.h"

namespace vg {

namespace {

struct FakeFrame {
    std::thread worker;
    std::vector<DevMatch> found;          // every candidate of the dispatch, index order not guaranteed
    std::vector<uint8_t> dump;            // dump mode: payloads in index order
    std::vector<vgen_match> sorted;       // after rt_wait
};

struct FakeCtx : vgen_ctx {
    std::vector<FakeFrame> ff;
    std::vector<uint32_t> chk_lut, dfa_blob;   // the filter's tables (host copies the DevFilter points at)
    unsigned stream_delay_ms = 2;              // what creating a hardware queue costs, scaled down
};

FakeCtx *fc(vgen_ctx *c) { return static_cast<FakeCtx *>(c); }

// beta, the cube root of unity behind the endomorphism (the constant of kernels.hip: fe_set_beta)
void set_beta(fe &b) {
    const u32 n[9] = {0x119501EEu, 0x09CB6143u, 0x1D626570u, 0x0092EA25u, 0x034E99CFu, 0x03CF561Au, 0x1C41B991u, 0x056CAF80u, 0x007AE96Au};
    for (int i = 0; i < 9; i++) b.n[i] = n[i];
}

// payload words (memory order) of an affine public key, as kernels.hip: payload_from_point does for the 20-byte formats
void payload_words(uint32_t fmt, const fe &x, const fe &y, u32 out[8]) {
    u32 xw[8], yw[8];
    fe_to_words(x, xw);
    fe_to_words(y, yw);
    memset(out, 0, 32);
    if (fmt == VGF_P2PKH || fmt == VGF_P2WPKH) {
        u32 sha[8];
        sha256_pub33(2u | (y.n[0] & 1u), xw, sha);
        ripemd160_of_sha(sha, out);
    } else if (fmt == VGF_P2SH_P2WPKH) {
        u32 sha[8], h[5];
        sha256_pub33(2u | (y.n[0] & 1u), xw, sha);
        ripemd160_of_sha(sha, h);
        sha256_script22(h, sha);
        ripemd160_of_sha(sha, out);
    } else if (fmt == VGF_P2PKH_UNCOMPRESSED) {
        u32 sha[8];
        sha256_pub65(xw, yw, sha);
        ripemd160_of_sha(sha, out);
    } else {
        keccak256_pub64_addr(xw, yw, out);
    }
}

bool candidate(const FakeCtx *c, const u32 *pl) {
    const int nw = (int)c->payload_words;
    if (c->h_filter.kind == DEVF_DFA)
        return nw == 8 ? dfa_match_payload_n<8>(c->h_filter.dfa_blob, (int)c->format, pl) : dfa_match_payload_n<5>(c->h_filter.dfa_blob, (int)c->format, pl);
    return nw == 8 ? filter_eval_n<8>(&c->h_filter, pl) : filter_eval_n<5>(&c->h_filter, pl);
}

bool dump_mode(const vgen_ctx *c) { return !c->have_filter || c->h_filter.kind == DEVF_HOST_ALL; }

void emit(FakeCtx *c, FakeFrame &ff, bool dump, uint32_t index, const u32 *pl, bool ok) {
    const size_t pb = (size_t)c->payload_words * 4;
    if (dump) {
        if (ok) memcpy(ff.dump.data() + (size_t)index * pb, pl, pb);
        return;
    }
    if (!ok || !candidate(c, pl)) return;
    DevMatch m;
    m.index = index;
    m.reserved = 0;
    memset(m.payload, 0, sizeof m.payload);
    memcpy(m.payload, pl, pb);
    ff.found.push_back(m);
}

// one key through the host's own derivation (full multiplication): taproot, explicit and random scalars, batches at n
void emit_key(FakeCtx *c, FakeFrame &ff, bool dump, uint32_t index, const Scalar &k) {
    u32 pl[8] = {0};
    bool ok = scalar_is_valid(k);
    if (ok) {
        uint8_t kb[32], out[32];
        scalar_to_be(k, kb);
        ok = payload_from_key(c->format, kb, out) != 0;
        memcpy(pl, out, (size_t)c->payload_words * 4);
    }
    emit(c, ff, dump, index, pl, ok);
}

// a key of the arbitrary-scalar path: itself, or (endomorphism contexts) its six images at variant * batch + i — one
// multiplication, then (beta^e x, +-y) as the kernels do
void emit_images(FakeCtx *c, FakeFrame &ff, bool dump, uint32_t i, const Scalar &k, bool endo) {
    if (!endo) {
        emit_key(c, ff, dump, i, k);
        return;
    }
    ge pt;
    const bool ok = scalar_is_valid(k) && host_ec_mul_gen(k, pt);
    fe x = pt.x, y = pt.y, beta;
    if (ok) {
        fe_normalize(x);
        fe_normalize(y);
    }
    set_beta(beta);
    for (uint32_t e = 0; e < 3; e++) {
        if (ok && e) {
            fe_mul(x, x, beta);
            fe_normalize(x);
        }
        for (uint32_t sgn = 0; sgn < 2; sgn++) {
            u32 pl[8] = {0};
            if (ok) {
                fe yy = y;
                if (sgn) {
                    fe_neg(yy, y, 1);
                    fe_normalize(yy);
                }
                payload_words(c->format, x, yy, pl);
            }
            emit(c, ff, dump, (sgn * 3 + e) * c->batch + i, pl, ok);
        }
    }
}

// the sequential walk k0 + i: one mixed addition per key, affine conversion in chunks sharing an inversion
void walk(FakeCtx *c, FakeFrame &ff, bool dump, Scalar k0, bool endo) {
    const uint32_t N = c->batch;
    if (c->format == VGF_P2TR || scalar_distance_to_n(k0) <= (uint64_t)N + 16) {
        for (uint32_t i = 0; i < N; i++) {
            Scalar k;
            if (scalar_add_u64(k, k0, i)) memset(&k, 0, sizeof k);
            emit_key(c, ff, dump, i, k);
        }
        return;
    }
    ge g, base;
    ge_generator(g);
    if (!host_ec_mul_gen(k0, base)) return;
    gej acc;
    gej_from_ge(acc, base);
    constexpr uint32_t CH = 512;
    std::vector<gej> jac(CH);
    std::vector<ge> aff(CH);
    fe beta;
    set_beta(beta);
    for (uint32_t i0 = 0; i0 < N; i0 += CH) {
        const uint32_t n = std::min(CH, N - i0);
        for (uint32_t j = 0; j < n; j++) {
            jac[j] = acc;
            gej nx;
            gej_add_ge(nx, acc, g);
            acc = nx;
        }
        host_batch_to_affine(jac.data(), aff.data(), n);
        for (uint32_t j = 0; j < n; j++) {
            u32 pl[8];
            fe x = aff[j].x, y = aff[j].y;
            fe_normalize(x);
            fe_normalize(y);
            if (!endo) {
                payload_words(c->format, x, y, pl);
                emit(c, ff, dump, i0 + j, pl, true);
                continue;
            }
            for (uint32_t e = 0; e < 3; e++) {          // images e = power of beta, s = negated (kernels.hip: ENDO)
                if (e) {
                    fe_mul(x, x, beta);
                    fe_normalize(x);
                }
                for (uint32_t sgn = 0; sgn < 2; sgn++) {
                    fe yy = y;
                    if (sgn) {
                        fe_neg(yy, y, 1);
                        fe_normalize(yy);
                    }
                    payload_words(c->format, x, yy, pl);
                    emit(c, ff, dump, (sgn * 3 + e) * N + i0 + j, pl, true);
                }
            }
        }
    }
}

int start_dispatch(FakeCtx *c, uint32_t frame, uint64_t keys, std::function<void(FakeFrame &, bool)> body, bool endo = false) {
    vgen_ctx::Frame &f = c->fr[frame];
    FakeFrame &ff = c->ff[frame];
    const bool dump = dump_mode(c);
    if (dump && c->dump_frames && frame >= c->dump_frames)
        return c->fail(VGEN_E_STATE, "dump mode serves frames 0.." + std::to_string(c->dump_frames - 1) + " of this context (pinned-memory budget)");
    ff.found.clear();
    if (dump) {
        ff.dump.assign((size_t)c->batch * (endo || keys > c->batch ? 6 : 1) * c->payload_words * 4, 0);
        if (!c->dump_frames) c->dump_frames = c->frames;
    }
    f.in_flight = true;
    f.dumped = dump;
    f.keys_tested = keys;
    f.endo_applied = endo || keys > c->batch;
    ff.worker = std::thread([body, &ff, dump]() { body(ff, dump); });
    return VGEN_OK;
}

}  // namespace

int rt_device_count(int *n, std::string &) {
    *n = 1;
    return VGEN_OK;
}

int rt_device_name(int device, std::string &name, std::string &err) {
    if (device != 0) {
        err = "fake runtime: one device";
        return VGEN_E_NODEVICE;
    }
    name = "fake device (CPU stand-in for the sanitizer builds)";
    return VGEN_OK;
}

int rt_create(const vgen_params *p_in, vgen_ctx **out, std::string &err) {
    if (!p_in || !out || (p_in->struct_size != sizeof(vgen_params) && p_in->struct_size != 28)) {   // ABI 4's block or ABI 3's 28 bytes
        err = "vgen_create: bad parameter block";
        return VGEN_E_INVALID;
    }
    vgen_params p_full;
    memset(&p_full, 0, sizeof p_full);
    memcpy(&p_full, p_in, p_in->struct_size);
    const vgen_params *p = &p_full;
    if (p->device != 0) {
        err = "device index out of range";
        return VGEN_E_NODEVICE;
    }
    if (p->format > VGF_ETHEREUM) {
        err = "unknown address format";
        return VGEN_E_INVALID;
    }
    FakeCtx *c = new FakeCtx();
    c->device = 0;
    c->batch = p->batch_size ? p->batch_size : (1u << 20);
    c->frames = p->frames ? p->frames : 12;
    c->match_cap = p->match_cap ? p->match_cap : 4096;
    c->format = p->format;
    c->payload_words = p->format == VGF_P2TR ? 8 : 5;
    c->endo = (p->flags & VGEN_FLAG_ENDO) != 0 && p->format != VGF_P2TR;
    c->S = 8;
    if (c->frames > 20 || c->batch % 8192 != 0 || c->batch > VGEN_MAX_BATCH) {
        err = "frames must be <= 20 and batch_size a multiple of 8192, at most 2^24";
        delete c;
        return VGEN_E_INVALID;
    }
    if (c->match_cap < 256) c->match_cap = 256;
    c->fr.resize(c->frames);
    c->ff.resize(c->frames);
    c->streams.assign(c->frames, nullptr);
    c->claimed.assign(c->frames, 0);
    c->streams[0] = reinterpret_cast<hipStream_t>(1);   // a context starts with frame 0's queue, as the real one does
    *out = c;
    return VGEN_OK;
}

void rt_destroy(vgen_ctx *c0) {
    if (!c0) return;
    FakeCtx *c = fc(c0);
    c->maker_cancel.store(true);
    if (c->stream_maker.joinable()) c->stream_maker.join();
    for (auto &ff : c->ff)
        if (ff.worker.joinable()) ff.worker.join();
    delete c;
}

// the helper thread of runtime.cpp: "creates" the remaining streams while the scan runs on the frames it has
bool rt_prepare_streams(vgen_ctx *c0) {
    FakeCtx *c = fc(c0);
    std::lock_guard<std::mutex> g(c->stream_mu);
    if (c->maker_started) return true;
    c->maker_started = true;
    c->stream_maker = std::thread([c]() {
        for (uint32_t i = 0; i < c->streams.size(); i++) {
            if (c->maker_cancel.load()) return;
            {
                std::lock_guard<std::mutex> lk(c->stream_mu);
                if (c->streams[i] || c->claimed[i]) continue;
                c->claimed[i] = 1;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(c->stream_delay_ms));
            {
                std::lock_guard<std::mutex> lk(c->stream_mu);
                c->claimed[i] = 0;
                c->streams[i] = reinterpret_cast<hipStream_t>(1);
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

static int ensure_frame(FakeCtx *c, uint32_t frame) {
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.s) return VGEN_OK;
    std::unique_lock<std::mutex> lk(c->stream_mu);
    c->stream_cv.wait(lk, [&]() { return c->streams[frame] || !c->claimed[frame]; });
    if (!c->streams[frame]) {
        c->claimed[frame] = 1;
        lk.unlock();
        std::this_thread::sleep_for(std::chrono::milliseconds(c->stream_delay_ms));
        lk.lock();
        c->claimed[frame] = 0;
        c->streams[frame] = reinterpret_cast<hipStream_t>(1);
        c->stream_cv.notify_all();
    }
    f.s = c->streams[frame];
    return VGEN_OK;
}

int rt_set_match_cap(vgen_ctx *c, uint32_t cap) {
    for (auto &f : c->fr)
        if (f.in_flight) return c->fail(VGEN_E_STATE, "vgen_set_match_cap while a dispatch is in flight");
    if (cap < 256) cap = 256;
    const uint64_t most = (uint64_t)c->batch * (c->endo ? 6 : 1);
    if (cap > most) cap = (uint32_t)most;
    c->match_cap = cap;
    return VGEN_OK;
}

int rt_set_filter(vgen_ctx *c0, const vgen_filter *f) {
    FakeCtx *c = fc(c0);
    for (auto &fr : c->fr)
        if (fr.in_flight) return c->fail(VGEN_E_STATE, "vgen_set_filter while a dispatch is in flight");
    if (!f) {
        c->have_filter = false;
        return VGEN_OK;
    }
    if (f->format != c->format) return c->fail(VGEN_E_INVALID, "filter was compiled for another address format");
    c->h_filter = f->dev;
    c->chk_lut = f->chk_lut;
    c->dfa_blob = f->dfa_blob;
    if (f->dev.chk_lut) c->h_filter.chk_lut = c->chk_lut.data();
    if (f->dev.kind == DEVF_DFA) c->h_filter.dfa_blob = c->dfa_blob.data();
    c->have_filter = true;
    return VGEN_OK;
}

int rt_dispatch(vgen_ctx *c0, uint32_t frame, const uint8_t start_key_be[32]) {
    FakeCtx *c = fc(c0);
    if (frame >= c->frames || !start_key_be) return c->fail(VGEN_E_INVALID, "bad frame index / key");
    if (c->fr[frame].in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    if (c->injected_fault()) return c->fail(VGEN_E_HIP, "injected device failure (vgen_debug_fail_after)");
    Scalar k0;
    scalar_from_be(k0, start_key_be);
    if (!scalar_is_valid(k0)) return c->fail(VGEN_E_RANGE, "start key is not a valid secp256k1 scalar");
    if (int rc = ensure_frame(c, frame)) return rc;
    c->fr[frame].start = k0;
    const bool endo = c->endo;
    return start_dispatch(c, frame, endo ? (uint64_t)c->batch * 6 : c->batch, [c, k0, endo](FakeFrame &ff, bool dump) { walk(c, ff, dump, k0, endo); });
}

int rt_dispatch_keys(vgen_ctx *c0, uint32_t frame, const uint8_t *keys_be, uint32_t n) {
    FakeCtx *c = fc(c0);
    if (frame >= c->frames || !keys_be) return c->fail(VGEN_E_INVALID, "bad frame index / key buffer");
    if (n == 0 || n > c->batch) return c->fail(VGEN_E_INVALID, "vgen_dispatch_keys: n must be in [1, batch_size]");
    if (c->fr[frame].in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    if (c->injected_fault()) return c->fail(VGEN_E_HIP, "injected device failure (vgen_debug_fail_after)");
    if (int rc = ensure_frame(c, frame)) return rc;
    std::vector<uint8_t> keys(keys_be, keys_be + (size_t)n * 32);
    const bool endo = c->endo;
    return start_dispatch(c, frame, endo ? (uint64_t)n * 6 : n, [c, keys, n, endo](FakeFrame &ff, bool dump) {
        for (uint32_t i = 0; i < n; i++) {
            Scalar k;
            scalar_from_be(k, keys.data() + (size_t)i * 32);
            emit_images(c, ff, dump, i, k, endo);
        }
    }, endo);
}

int rt_dispatch_random(vgen_ctx *c0, uint32_t frame, const RndSeed &seed, uint32_t stream, uint64_t first_index) {
    FakeCtx *c = fc(c0);
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    if (first_index > UINT64_MAX - (c->batch - 1)) return c->fail(VGEN_E_RANGE, "vgen_dispatch_random: index range wraps 2^64");
    if (c->fr[frame].in_flight) return c->fail(VGEN_E_STATE, "frame already has a dispatch in flight");
    if (c->injected_fault()) return c->fail(VGEN_E_HIP, "injected device failure (vgen_debug_fail_after)");
    if (int rc = ensure_frame(c, frame)) return rc;
    const bool endo = c->endo;
    return start_dispatch(c, frame, endo ? (uint64_t)c->batch * 6 : c->batch, [c, seed, stream, first_index, endo](FakeFrame &ff, bool dump) {
        for (uint32_t i = 0; i < c->batch; i++) {
            Scalar k;
            const uint64_t idx = first_index + i;
            rnd_scalar(seed, stream, (uint32_t)idx, (uint32_t)(idx >> 32), k.w);
            emit_images(c, ff, dump, i, k, endo);
        }
    }, endo);
}

int rt_wait(vgen_ctx *c0, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches, uint64_t *keys_tested) {
    FakeCtx *c = fc(c0);
    if (frame >= c->frames) return c->fail(VGEN_E_INVALID, "bad frame index");
    vgen_ctx::Frame &f = c->fr[frame];
    if (!f.in_flight) return c->fail(VGEN_E_STATE, "No pending operation on frame " + std::to_string(frame));
    FakeFrame &ff = c->ff[frame];
    if (ff.worker.joinable()) ff.worker.join();
    f.in_flight = false;
    if (keys_tested) *keys_tested = f.keys_tested;
    uint32_t found = 0;
    if (!f.dumped) {
        found = (uint32_t)ff.found.size();
        // the ring keeps the first match_cap records in ARRIVAL order (which lanes arrive first is not defined on the
        // device either); what is kept is then sorted by index
        const uint32_t stored = std::min(found, c->match_cap);
        std::sort(ff.found.begin(), ff.found.begin() + stored, [](const DevMatch &x, const DevMatch &y) { return x.index < y.index; });
        if (out) {
            const uint32_t n = std::min(stored, cap);
            for (uint32_t i = 0; i < n; i++) {
                out[i].index = ff.found[i].index;
                out[i].reserved = 0;
                memcpy(out[i].payload, ff.found[i].payload, 32);
            }
        }
    }
    if (n_matches) *n_matches = found;
    return VGEN_OK;
}

int rt_dump_view(vgen_ctx *c0, uint32_t frame, const uint8_t **ptr, size_t *len) {
    FakeCtx *c = fc(c0);
    if (frame >= c->frames || !ptr) return c->fail(VGEN_E_INVALID, "bad frame index / pointer");
    vgen_ctx::Frame &f = c->fr[frame];
    if (f.in_flight) return c->fail(VGEN_E_STATE, "vgen_read_dump before vgen_wait");
    if (!f.dumped) return c->fail(VGEN_E_STATE, "frame's last dispatch was not in dump mode");
    *ptr = c->ff[frame].dump.data();
    if (len) *len = c->ff[frame].dump.size();
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

int rt_frame_times(vgen_ctx *c, uint32_t, float *, float *) { return c->fail(VGEN_E_STATE, "fake runtime: no event timing"); }
int rt_frame_clock(vgen_ctx *, uint32_t, uint32_t *cycles, uint32_t *ticks) {
    if (cycles) *cycles = 0;
    if (ticks) *ticks = 0;
    return VGEN_OK;
}
int rt_clock_probe_start(vgen_ctx *c, uint32_t) { return c->fail(VGEN_E_UNSUPPORTED, "fake runtime: no clock probe"); }
int rt_clock_probe_read(vgen_ctx *c, double *) { return c->fail(VGEN_E_UNSUPPORTED, "fake runtime: no clock probe"); }
void rt_prefer_table_bits(vgen_ctx *c, uint32_t bits, uint32_t cap) {   // (the stand-in multiplies on the host: nothing to build)
    c->gtab_bits_pref = bits;
    c->gtab_bits_cap = cap;
}
int rt_get_memory(const vgen_ctx *c, vgen_memory_info *out) {
    if (!out || out->struct_size != sizeof(vgen_memory_info)) return VGEN_E_INVALID;
    const uint32_t sz = out->struct_size;
    memset(out, 0, sizeof *out);
    out->struct_size = sz;
    out->budget_bytes = c->mem_budget;
    return VGEN_OK;
}
int rt_get_resources(const vgen_ctx *c, uint32_t *dump_frames, uint32_t *table_bits, uint32_t *table_bits_wanted, std::string *note) {
    if (dump_frames) *dump_frames = c->dump_frames ? c->dump_frames : c->frames;
    if (table_bits) *table_bits = 0;      // (the stand-in multiplies on the host: no generator table)
    if (table_bits_wanted) *table_bits_wanted = 0;
    if (note) note->clear();
    return VGEN_OK;
}

}  // namespace vg
