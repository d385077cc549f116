// cabi.cpp — the extern "C" surface declared in include/vgen_hip.h.
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/vgen_hip.h"
#include "host/encode.h"
#include "host/filter.h"
#include "host/pattern_info.h"
#include "host/provider.h"
#include "host/scalar.h"
#include "runtime.h"

namespace {
thread_local std::string g_last_error;

// (No load-time side effects: the library neither sets nor needs GPU_MAX_HW_QUEUES.  A context drives every frame
// through a stream of its own, spread over the runtime's stream priority levels so that each owns a hardware queue —
// runtime.cpp: create_stream.)

int copy_out(const std::string &s, char *out, size_t cap) {
    if (!out || cap < s.size() + 1) return VGEN_E_INVALID;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
}  // namespace

extern "C" {

int vgen_abi_version(void) { return VGEN_ABI_VERSION; }

int vgen_device_count(int *n) {
    if (!n) return VGEN_E_INVALID;
    return vg::rt_device_count(n, g_last_error);
}

int vgen_device_name(int device, char *buf, size_t cap) {
    if (!buf || cap == 0) return VGEN_E_INVALID;
    std::string s;
    if (int rc = vg::rt_device_name(device, s, g_last_error)) return rc;
    strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return VGEN_OK;
}

int vgen_create(const vgen_params *p, vgen_ctx **out) {
    std::string err;
    int rc = vg::rt_create(p, out, err);
    if (rc != VGEN_OK) g_last_error = err;
    return rc;
}

void vgen_destroy(vgen_ctx *ctx) { vg::rt_destroy(ctx); }

const char *vgen_last_error(const vgen_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int vgen_get_info(const vgen_ctx *ctx, uint32_t *batch_size, uint32_t *frames, uint32_t *match_cap) {
    if (!ctx) return VGEN_E_INVALID;
    if (batch_size) *batch_size = ctx->batch;
    if (frames) *frames = ctx->frames;
    if (match_cap) *match_cap = ctx->match_cap;
    return VGEN_OK;
}

int vgen_get_resources(const vgen_ctx *ctx, uint32_t *dump_frames, uint32_t *table_bits, uint32_t *table_bits_wanted, char *note, size_t note_cap) {
    if (!ctx) return VGEN_E_INVALID;
    std::string why;
    const int rc = vg::rt_get_resources(ctx, dump_frames, table_bits, table_bits_wanted, &why);
    if (note && note_cap) {
        strncpy(note, why.c_str(), note_cap - 1);
        note[note_cap - 1] = 0;
    }
    return rc;
}

int vgen_get_memory(const vgen_ctx *ctx, vgen_memory_info *out) {
    if (!ctx || !out) return VGEN_E_INVALID;
    return vg::rt_get_memory(ctx, out);
}

int vgen_filter_compile(const char *pattern, int case_insensitive, uint32_t format, vgen_filter **out) {
    if (!pattern || !out) return VGEN_E_INVALID;
    vgen_filter *f = new vgen_filter();
    std::string err;
    if (!vg::filter_compile(pattern, case_insensitive != 0, format, *f, err)) {
        g_last_error = err;
        delete f;
        return VGEN_E_PATTERN;
    }
    *out = f;
    return VGEN_OK;
}

void vgen_filter_free(vgen_filter *f) { delete f; }

int vgen_filter_matches(const vgen_filter *f, const char *address) {
    if (!f || !address) return VGEN_E_INVALID;
    return f->dfa.is_match(address) ? 1 : 0;
}

int vgen_filter_device_kind(const vgen_filter *f) { return f ? (int)f->dev.kind : VGEN_E_INVALID; }

int vgen_filter_dfa_bytes(const vgen_filter *f) { return f ? (f->dev.kind == vg::DEVF_DFA ? (int)f->dev.dfa_bytes : 0) : VGEN_E_INVALID; }

int vgen_pattern_invalid_chars(const char *pattern, int case_insensitive, uint32_t format, char *out, size_t cap,
                               size_t *n) {
    if (!pattern || !n || !vg::format_charset_name(format)) return VGEN_E_INVALID;
    const std::string bad = vg::pattern_invalid_chars(pattern, case_insensitive != 0, format);
    *n = bad.size();
    if (out && cap) {
        const size_t m = bad.size() < cap - 1 ? bad.size() : cap - 1;
        memcpy(out, bad.data(), m);
        out[m] = 0;
    }
    return VGEN_OK;
}

int vgen_pattern_difficulty(const char *pattern, int case_insensitive, uint32_t format, uint64_t *out) {
    if (!pattern || !out || !vg::format_charset_name(format)) return VGEN_E_INVALID;
    *out = vg::pattern_difficulty(pattern, case_insensitive != 0, format);
    return VGEN_OK;
}

const char *vgen_format_charset_name(uint32_t format) { return vg::format_charset_name(format); }

int vgen_provider_resolve(const char *pattern, const char *table_path, char *address, size_t acap, uint32_t *format,
                          int32_t *has_range, uint8_t start_be[32], uint8_t end_be[32]) {
    if (!pattern || !address || !format || !has_range || !start_be || !end_be) return VGEN_E_INVALID;
    vg::ProviderResult r;
    std::string err;
    const int rc = vg::provider_resolve(pattern, table_path, r, err);
    if (rc < 0) {
        g_last_error = err;
        return VGEN_E_INVALID;
    }
    if (rc == 0) return 0;
    if (r.address.size() + 1 > acap) return VGEN_E_INVALID;
    memcpy(address, r.address.c_str(), r.address.size() + 1);
    *format = r.format;
    *has_range = r.has_range ? 1 : 0;
    memcpy(start_be, r.start, 32);
    memcpy(end_be, r.end, 32);
    return 1;
}

int vgen_provider_build_pattern(const char *address, uint32_t prefix_length, char *out, size_t cap) {
    if (!address || !out) return VGEN_E_INVALID;
    const std::string p = prefix_length ? vg::provider_build_pattern(address, prefix_length) : vg::provider_build_exact_pattern(address);
    if (p.size() + 1 > cap) return VGEN_E_INVALID;
    memcpy(out, p.c_str(), p.size() + 1);
    return VGEN_OK;
}

int vgen_clock_probe_start(vgen_ctx *ctx, uint32_t duration_ms) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_clock_probe_start(ctx, duration_ms);
}

int vgen_clock_probe_read(vgen_ctx *ctx, double *mhz) {
    if (!ctx || !mhz) return VGEN_E_INVALID;
    return vg::rt_clock_probe_read(ctx, mhz);
}

int vgen_set_match_cap(vgen_ctx *ctx, uint32_t match_cap) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_set_match_cap(ctx, match_cap);
}

int vgen_set_filter(vgen_ctx *ctx, const vgen_filter *f) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_set_filter(ctx, f);
}

int vgen_dispatch(vgen_ctx *ctx, uint32_t frame, const uint8_t start_key_be[32]) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_dispatch(ctx, frame, start_key_be);
}

int vgen_dispatch_keys(vgen_ctx *ctx, uint32_t frame, const uint8_t *keys_be, uint32_t n) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_dispatch_keys(ctx, frame, keys_be, n);
}

#ifdef VGEN_TEST_HOOKS
// Fault injection: only in the test build of the library (tests/native/libvgen_hip_hooks.so, tests/native/vgen_hip_hooks.h).
extern "C" int vgen_debug_fail_after(vgen_ctx *ctx, uint64_t after_dispatches) {
    if (!ctx) return VGEN_E_INVALID;
    ctx->fail_after = after_dispatches;
    return VGEN_OK;
}
#endif

int vgen_dispatch_random(vgen_ctx *ctx, uint32_t frame, uint64_t seed, uint32_t stream, uint64_t first_index) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_dispatch_random(ctx, frame, vg::rnd_seed_from_u64(seed), stream, first_index);
}

int vgen_dispatch_random_seed(vgen_ctx *ctx, uint32_t frame, const uint8_t seed[24], uint32_t stream, uint64_t first_index) {
    if (!ctx || !seed) return VGEN_E_INVALID;
    return vg::rt_dispatch_random(ctx, frame, vg::rnd_seed_from_bytes(seed), stream, first_index);
}

int vgen_random_key(uint64_t seed, uint32_t stream, uint64_t index, uint8_t key_be[32]) {
    if (!key_be) return VGEN_E_INVALID;
    return vg::random_key_be(vg::rnd_seed_from_u64(seed), stream, index, key_be) ? VGEN_OK : VGEN_E_RANGE;
}

int vgen_random_key_seed(const uint8_t seed[24], uint32_t stream, uint64_t index, uint8_t key_be[32]) {
    if (!seed || !key_be) return VGEN_E_INVALID;
    return vg::random_key_be(vg::rnd_seed_from_bytes(seed), stream, index, key_be) ? VGEN_OK : VGEN_E_RANGE;
}

int vgen_wait(vgen_ctx *ctx, uint32_t frame, vgen_match *out, uint32_t cap, uint32_t *n_matches,
              uint64_t *keys_tested) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_wait(ctx, frame, out, cap, n_matches, keys_tested);
}

int vgen_read_dump(vgen_ctx *ctx, uint32_t frame, uint8_t *out, size_t out_len) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_read_dump(ctx, frame, out, out_len);
}

int vgen_dump_view(vgen_ctx *ctx, uint32_t frame, const uint8_t **ptr, size_t *len) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_dump_view(ctx, frame, ptr, len);
}

int vgen_get_topology(const vgen_ctx *ctx, uint32_t *streams, uint32_t *hw_queues, uint32_t *priority_levels,
                      int32_t *oversubscribed) {
    if (!ctx) return VGEN_E_INVALID;
    if (streams) *streams = ctx->frames;
    if (hw_queues) *hw_queues = ctx->hw_queues;
    if (priority_levels) *priority_levels = ctx->prio_levels;
    if (oversubscribed) *oversubscribed = vg::rt_oversubscribed(ctx) ? 1 : 0;
    return VGEN_OK;
}

int vgen_frame_kernel_ms(vgen_ctx *ctx, uint32_t frame, float *ms) {
    if (!ctx || !ms) return VGEN_E_INVALID;
    return vg::rt_frame_times(ctx, frame, ms, nullptr);
}

int vgen_frame_clock(vgen_ctx *ctx, uint32_t frame, uint32_t *cycles, uint32_t *ticks_100mhz) {
    if (!ctx) return VGEN_E_INVALID;
    return vg::rt_frame_clock(ctx, frame, cycles, ticks_100mhz);
}

int vgen_frame_dispatch_ms(vgen_ctx *ctx, uint32_t frame, float *ms) {
    if (!ctx || !ms) return VGEN_E_INVALID;
    return vg::rt_frame_times(ctx, frame, nullptr, ms);
}

int vgen_address_from_payload(uint32_t format, const uint8_t *payload, char *out, size_t cap) {
    if (!payload) return VGEN_E_INVALID;
    std::string s = vg::address_from_payload(format, payload);
    if (s.empty()) return VGEN_E_UNSUPPORTED;
    return copy_out(s, out, cap);
}

int vgen_key_to_wif(uint32_t format, const uint8_t key_be[32], char *out, size_t cap) {
    if (!key_be) return VGEN_E_INVALID;
    vg::Scalar k;
    vg::scalar_from_be(k, key_be);
    if (!vg::scalar_is_valid(k)) return VGEN_E_RANGE;
    return copy_out(vg::key_to_wif(format, key_be), out, cap);
}

int vgen_key_add(const uint8_t key_be[32], uint64_t amount, uint8_t out_be[32]) {
    if (!key_be || !out_be) return VGEN_E_INVALID;
    vg::Scalar k, r;
    vg::scalar_from_be(k, key_be);
    uint32_t carry = vg::scalar_add_u64(r, k, amount);
    vg::scalar_to_be(r, out_be);
    return (carry || !vg::scalar_is_valid(r)) ? VGEN_E_RANGE : VGEN_OK;
}

int vgen_key_variant(const uint8_t key_be[32], uint32_t variant, uint8_t out_be[32]) {
    if (!key_be || !out_be || variant >= 6) return VGEN_E_INVALID;
    vg::Scalar k, r;
    vg::scalar_from_be(k, key_be);
    if (!vg::scalar_is_valid(k)) return VGEN_E_RANGE;
    vg::scalar_variant(r, k, variant);
    vg::scalar_to_be(r, out_be);
    return VGEN_OK;
}

int vgen_derive(uint32_t format, const uint8_t key_be[32], char *address, size_t acap, char *wif, size_t wcap) {
    if (!key_be) return VGEN_E_INVALID;
    uint8_t payload[32];
    int n = vg::payload_from_key(format, key_be, payload);
    if (n == 0) {
        vg::Scalar k;
        vg::scalar_from_be(k, key_be);
        return vg::scalar_is_valid(k) ? VGEN_E_UNSUPPORTED : VGEN_E_RANGE;
    }
    if (address) {
        int rc = copy_out(vg::address_from_payload(format, payload), address, acap);
        if (rc < 0) return rc;
    }
    if (wif) {
        int rc = copy_out(vg::key_to_wif(format, key_be), wif, wcap);
        if (rc < 0) return rc;
    }
    return VGEN_OK;
}

}  // extern "C"
