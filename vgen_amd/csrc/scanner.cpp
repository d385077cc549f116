// scanner.cpp — vgen_scan: the host loop of the GPU scan, mirroring scan_gpu_with_runner
// (reference src/gpu.rs:920-1125) above the C ABI.
//
// Same control flow as the reference: pick the base key (config.start, or a random valid scalar —
// gpu.rs:933-945), prime every frame (gpu.rs:973-995), then round-robin: await a frame, immediately
// re-dispatch it with the next batch (gpu.rs:1003-1028), turn that frame's results into matches
// (gpu.rs:1030-1104), add batch_size to the operation count and call the progress callback
// (gpu.rs:1106-1109); stop when `count` matches exist and nothing more was dispatched (gpu.rs:1111).
// Differences, all on the host side of the boundary: candidates arrive pre-filtered by the device and
// are confirmed with the exact DFA (the reference encodes and regex-matches all batch_size hashes on
// rayon); the base key can be seeded; batches can be striped over several contexts (multi-GPU).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../include/vgen_hip.h"
#include "host/encode.h"
#include "host/filter.h"
#include "host/scalar.h"
#include "runtime.h"

namespace vg {

// k0(seed, shard) = SHA-256("vgen-mi355x" || u64le(seed) || u32le(shard)) mod n, re-drawn if 0
// (BASELINE.md §4).
void seed_key(uint64_t seed, uint32_t shard, Scalar &out) {
    uint32_t redraw = 0;
    for (;;) {
        uint8_t buf[11 + 8 + 4 + 4];
        size_t n = 11;
        memcpy(buf, "vgen-mi355x", 11);
        for (int i = 0; i < 8; i++) buf[n++] = (uint8_t)(seed >> (8 * i));
        for (int i = 0; i < 4; i++) buf[n++] = (uint8_t)(shard >> (8 * i));
        if (redraw)
            for (int i = 0; i < 4; i++) buf[n++] = (uint8_t)(redraw >> (8 * i));
        uint8_t d[32];
        host_sha256(buf, n, d);
        scalar_from_be(out, d);
        if (scalar_cmp_words(out.w, SCALAR_N) >= 0) {
            // digest < 2^256 < 2n: one subtraction reduces it
            int64_t b = 0;
            for (int i = 0; i < 8; i++) {
                int64_t t = (int64_t)out.w[i] - SCALAR_N[i] + b;
                out.w[i] = (uint32_t)t;
                b = t >> 32;
            }
        }
        if (!scalar_is_zero(out)) return;
        redraw++;
    }
}

namespace {

struct Pending {
    Scalar start;
    bool valid = false;
};

void random_valid_key(Scalar &k) {
    std::random_device rd;
    for (;;) {
        for (int i = 0; i < 8; i++) k.w[i] = rd();
        if (scalar_is_valid(k)) return;   // rejection sampling, gpu.rs:938-944
    }
}

bool make_match(const vgen_filter &flt, uint32_t format, const Scalar &batch_start, uint32_t index,
                const uint8_t *payload, const Scalar *end, vgen_generated &g) {
    std::string addr = address_from_payload(format, payload);
    if (addr.empty() || !flt.dfa.is_match(addr)) return false;          // pattern.matches, gpu.rs:1069
    Scalar k;
    if (scalar_add_u64(k, batch_start, index) || !scalar_is_valid(k)) return false;   // increment_key -> None
    if (end && scalar_cmp(k, *end) > 0) return false;                   // gpu.rs:1074-1078
    uint8_t kb[32];
    scalar_to_be(k, kb);
    memset(&g, 0, sizeof g);
    std::string wif = key_to_wif(format, kb), hex = hex_lower(kb, 32);
    strncpy(g.address, addr.c_str(), sizeof g.address - 1);
    strncpy(g.wif, wif.c_str(), sizeof g.wif - 1);
    strncpy(g.hex, hex.c_str(), sizeof g.hex - 1);
    g.format = format;
    memcpy(g.key, kb, 32);
    return true;
}

}  // namespace
}  // namespace vg

using namespace vg;

namespace {

// One shard of a scan on one context.  `shared_found` (optional) is the match counter shared by the
// shards of a multi-device scan; without it the shard counts its own matches.
int scan_shard(vgen_ctx *ctx, const vgen_filter &flt, const vgen_scan_config *cfg, vgen_progress_cb cb, void *user,
               volatile int32_t *stop, std::atomic<uint64_t> *shared_found, std::atomic<uint64_t> *shared_ops,
               std::vector<vgen_generated> &matches, uint64_t &total_ops) {
    if (cfg->format != ctx->format) return ctx->fail(VGEN_E_INVALID, "scan format differs from the context's format");

    const uint32_t N = ctx->batch;
    const size_t pbytes = (size_t)ctx->payload_words * 4;
    const uint32_t shards = cfg->n_shards > 1 ? cfg->n_shards : 1;
    const uint32_t shard = cfg->n_shards > 1 ? cfg->shard : 0;
    if (shard >= shards) return ctx->fail(VGEN_E_INVALID, "shard >= n_shards");

    // device prefilter or host filtering of full dumps: too permissive a prefilter would overflow the
    // match ring, so it is only used when a batch is expected to produce few candidates
    // (a DEVF_DFA filter has no selectivity estimate: it starts on the device and falls back on overflow)
    bool host_all = flt.dev.kind == DEVF_HOST_ALL ||
                    (flt.dev.kind != DEVF_HOST_ALL && flt.selectivity * (double)N > (double)ctx->match_cap / 8);
    int rc = vgen_set_filter(ctx, host_all ? nullptr : &flt);
    if (rc != VGEN_OK) return rc;

    Scalar current;
    if (cfg->has_start) {
        scalar_from_be(current, cfg->start);
    } else if (cfg->seed) {
        seed_key(cfg->seed, 0, current);   // one base; shards stripe it (deterministic multi-GPU)
    } else {
        random_valid_key(current);
    }
    if (!scalar_is_valid(current)) return ctx->fail(VGEN_E_RANGE, "start key is not a valid secp256k1 scalar");
    Scalar end_key;
    const Scalar *end = nullptr;
    if (cfg->has_end) {
        scalar_from_be(end_key, cfg->end);
        end = &end_key;
    }
    // batch striping: this context takes global batches shard, shard + shards, ...
    bool exhausted = false;
    if (shard) exhausted = scalar_add_u64(current, current, (uint64_t)shard * N) || !scalar_is_valid(current);
    const uint64_t stride = (uint64_t)shards * N;

    uint64_t dispatched = 0;
    total_ops = 0;
    auto found = [&]() -> uint64_t { return shared_found ? shared_found->load(std::memory_order_relaxed) : matches.size(); };
    auto push = [&](const vgen_generated &g) {
        matches.push_back(g);
        if (shared_found) shared_found->fetch_add(1, std::memory_order_relaxed);
    };
    const uint64_t count = cfg->count;
    const uint32_t nf = ctx->frames;
    std::vector<Pending> pend(nf);
    std::vector<vgen_match> recs(ctx->match_cap);
    std::vector<uint8_t> dumpbuf;
    uint32_t in_flight = 0;
    int status = VGEN_OK;

    auto stopped = [&]() { return stop && *stop; };
    auto in_range = [&]() { return !exhausted && (!end || scalar_cmp(current, *end) <= 0); };
    auto can_dispatch = [&]() { return in_range() && (!cfg->max_batches || dispatched < cfg->max_batches); };
    auto dispatch = [&](uint32_t frame) -> int {
        uint8_t kb[32];
        scalar_to_be(current, kb);
        int r = vgen_dispatch(ctx, frame, kb);
        if (r != VGEN_OK) return r;
        pend[frame].start = current;
        pend[frame].valid = true;
        dispatched++;
        Scalar nx;
        if (scalar_add_u64(nx, current, stride) || !scalar_is_valid(nx)) exhausted = true;   // key space exhausted
        else current = nx;
        return VGEN_OK;
    };

    for (uint32_t i = 0; i < nf; i++) {
        if (!can_dispatch() || stopped() || found() >= count) break;
        if ((status = dispatch(i)) != VGEN_OK) break;
        in_flight++;
    }

    uint32_t frame = 0;
    while (status == VGEN_OK && in_flight > 0) {
        uint32_t n_found = 0;
        uint64_t tested = 0;
        if ((status = vgen_wait(ctx, frame, recs.data(), (uint32_t)recs.size(), &n_found, &tested)) != VGEN_OK) break;
        in_flight--;
        const Scalar batch_start = pend[frame].start;
        pend[frame].valid = false;
        const bool dumped = ctx->fr[frame].dumped;
        if (dumped) {
            dumpbuf.resize((size_t)N * pbytes);
            if ((status = vgen_read_dump(ctx, frame, dumpbuf.data(), dumpbuf.size())) != VGEN_OK) break;
        }

        bool dispatched_next = false;
        if (!stopped() && found() < count && can_dispatch()) {
            if ((status = dispatch(frame)) != VGEN_OK) break;
            in_flight++;
            dispatched_next = true;
        }

        if (dumped) {
            // host filtering of the whole batch (the reference's only mode, gpu.rs:1030-1093), in
            // parallel chunks, results kept in ascending index order
            unsigned nt = std::max(1u, std::thread::hardware_concurrency());
            std::vector<std::vector<vgen_generated>> part(nt);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    uint32_t lo = (uint32_t)((uint64_t)N * t / nt), hi = (uint32_t)((uint64_t)N * (t + 1) / nt);
                    vgen_generated g;
                    for (uint32_t i = lo; i < hi; i++)
                        if (make_match(flt, cfg->format, batch_start, i, &dumpbuf[(size_t)i * pbytes], end, g))
                            part[t].push_back(g);
                });
            for (auto &x : th) x.join();
            for (auto &p : part)
                for (auto &g : p)
                    if (found() < count) push(g);
        } else {
            if (n_found > recs.size()) {
                // More candidates than the ring holds (a permissive pattern): nothing may be dropped, so
                // drain what is in flight, switch to host filtering of full dumps (the reference's mode)
                // and redo from this batch.
                for (uint32_t f = 0; f < nf; f++)
                    if (ctx->fr[f].in_flight) (void)vgen_wait(ctx, f, nullptr, 0, nullptr, nullptr);
                if ((status = vgen_set_filter(ctx, nullptr)) != VGEN_OK) break;
                dispatched -= 1 + in_flight;
                in_flight = 0;
                current = batch_start;
                exhausted = false;
                for (uint32_t i = 0; i < nf; i++) {
                    if (!can_dispatch() || stopped() || found() >= count) break;
                    if ((status = dispatch(i)) != VGEN_OK) break;
                    in_flight++;
                }
                frame = 0;
                continue;
            }
            vgen_generated g;
            for (uint32_t i = 0; i < n_found && found() < count; i++)
                if (make_match(flt, cfg->format, batch_start, recs[i].index, recs[i].payload, end, g)) push(g);
        }

        total_ops += N;                              // gpu.rs:1106
        if (cb) cb(shared_ops ? shared_ops->fetch_add(N) + N : total_ops, user);
        if (found() >= count && !dispatched_next) break;   // gpu.rs:1111
        frame = (frame + 1) % nf;
    }
    // drain anything still in flight (the reference drops its runner; we must not leave frames busy)
    for (uint32_t f = 0; f < nf; f++)
        if (ctx->fr[f].in_flight) (void)vgen_wait(ctx, f, nullptr, 0, nullptr, nullptr);
    return status;
}

int finish_result(vgen_ctx *ctx, std::vector<vgen_generated> &matches, uint64_t ops, double secs, vgen_scan_result *out) {
    out->n_matches = matches.size();
    out->operations = ops;
    out->elapsed_secs = secs;
    if (!matches.empty()) {
        out->matches = (vgen_generated *)malloc(matches.size() * sizeof(vgen_generated));
        if (!out->matches) return ctx->fail(VGEN_E_NOMEM, "out of memory");
        memcpy(out->matches, matches.data(), matches.size() * sizeof(vgen_generated));
    }
    return VGEN_OK;
}

}  // namespace

extern "C" int vgen_scan(vgen_ctx *ctx, const char *pattern, const vgen_scan_config *cfg, vgen_progress_cb cb,
                         void *user, volatile int32_t *stop, vgen_scan_result *out) {
    if (!ctx || !pattern || !cfg || !out || cfg->struct_size != sizeof(vgen_scan_config)) return VGEN_E_INVALID;
    memset(out, 0, sizeof *out);
    const auto t0 = std::chrono::steady_clock::now();
    vgen_filter flt;
    std::string err;
    if (!filter_compile(pattern, cfg->case_insensitive != 0, cfg->format, flt, err))
        return ctx->fail(VGEN_E_PATTERN, err);
    std::vector<vgen_generated> matches;
    uint64_t ops = 0;
    int rc = scan_shard(ctx, flt, cfg, cb, user, stop, nullptr, nullptr, matches, ops);
    if (rc != VGEN_OK) return rc;
    return finish_result(ctx, matches, ops, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
}

// Multi-device scan: one host thread per context, batches striped over the contexts (context i takes
// global batches b = i mod n), a shared match counter and stop flag, results merged in ascending key
// order and truncated to `count` (no collective, host-side aggregation only).
extern "C" int vgen_scan_multi(vgen_ctx **ctxs, uint32_t n_ctx, const char *pattern, const vgen_scan_config *cfg,
                               vgen_progress_cb cb, void *user, volatile int32_t *stop, vgen_scan_result *out) {
    if (!ctxs || n_ctx == 0 || !pattern || !cfg || !out || cfg->struct_size != sizeof(vgen_scan_config)) return VGEN_E_INVALID;
    for (uint32_t i = 0; i < n_ctx; i++)
        if (!ctxs[i] || ctxs[i]->batch != ctxs[0]->batch) return VGEN_E_INVALID;
    memset(out, 0, sizeof *out);
    const auto t0 = std::chrono::steady_clock::now();
    vgen_filter flt;
    std::string err;
    if (!filter_compile(pattern, cfg->case_insensitive != 0, cfg->format, flt, err))
        return ctxs[0]->fail(VGEN_E_PATTERN, err);
    vgen_scan_config base = *cfg;
    if (!base.has_start) {   // all shards must walk the same base key
        Scalar k;
        if (base.seed) seed_key(base.seed, 0, k);
        else random_valid_key(k);
        scalar_to_be(k, base.start);
        base.has_start = 1;
    }
    std::atomic<uint64_t> found{0}, ops_shared{0};
    std::vector<std::vector<vgen_generated>> part(n_ctx);
    std::vector<uint64_t> ops(n_ctx, 0);
    std::vector<int> rcs(n_ctx, VGEN_OK);
    std::vector<std::thread> th;
    std::mutex cb_mu;
    struct CbCtx { vgen_progress_cb cb; void *user; std::mutex *mu; } cbc{cb, user, &cb_mu};
    auto locked_cb = [](uint64_t o, void *u) {
        CbCtx *c = (CbCtx *)u;
        std::lock_guard<std::mutex> g(*c->mu);
        c->cb(o, c->user);
    };
    for (uint32_t i = 0; i < n_ctx; i++)
        th.emplace_back([&, i]() {
            vgen_scan_config c = base;
            c.shard = i;
            c.n_shards = n_ctx;
            rcs[i] = scan_shard(ctxs[i], flt, &c, cb ? (vgen_progress_cb)locked_cb : nullptr, &cbc, stop, &found, &ops_shared,
                                part[i], ops[i]);
        });
    for (auto &x : th) x.join();
    for (uint32_t i = 0; i < n_ctx; i++)
        if (rcs[i] != VGEN_OK) return rcs[i];
    std::vector<vgen_generated> all;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_ctx; i++) {
        all.insert(all.end(), part[i].begin(), part[i].end());
        total += ops[i];
    }
    std::sort(all.begin(), all.end(), [](const vgen_generated &a, const vgen_generated &b) { return memcmp(a.key, b.key, 32) < 0; });
    if (all.size() > cfg->count) all.resize((size_t)cfg->count);
    return finish_result(ctxs[0], all, total, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
}

extern "C" void vgen_scan_result_free(vgen_scan_result *r) {
    if (!r) return;
    free(r->matches);
    memset(r, 0, sizeof *r);
}
