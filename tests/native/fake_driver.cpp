// fake_driver.cpp — TEST-ONLY: drives the host side of libvgen_hip.so (cabi.cpp + scanner.cpp + host/*.cpp, linked
// against the CPU stand-in of the runtime, fake_rt.cpp) through the C ABI and checks every result against the oracle.
// Built twice, with -fsanitize=thread and with -fsanitize=address,undefined (tests/native/Makefile), and run by
// tests/test_host_sanitizers.py in the build container: the scan loop's worker pool, helper-thread ramp, shared
// counters, checkpoint lock, multi-context threads and failure take-over under the sanitizers.
// usage: fake_driver [scenario ...]   (no argument: all);  exit status 0 = every check passed.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "vgen_hip_hooks.h"   // include/vgen_hip.h + the test build's fault-injection entry point
#include "../../oracle/vgen_oracle.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "CHECK FAILED %s:%d: %s  ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                                  \
            fprintf(stderr, "\n");                                         \
            g_fail++;                                                      \
        }                                                                  \
    } while (0)

static const uint32_t BATCH = 8192;
using Pairs = std::vector<std::pair<std::string, std::string>>;   // (address, wif)

static void key_of(uint64_t v, uint8_t be[32]) {
    memset(be, 0, 32);
    for (int i = 0; i < 8; i++) be[31 - i] = (uint8_t)(v >> (8 * i));
}

static vgen_ctx *make_ctx(uint32_t fmt, uint32_t frames, uint32_t cap = 0, uint32_t flags = 0) {
    vgen_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.batch_size = BATCH;
    p.format = fmt;
    p.frames = frames;
    p.match_cap = cap;
    p.flags = flags;
    vgen_ctx *c = nullptr;
    int rc = vgen_create(&p, &c);
    if (rc != VGEN_OK) {
        fprintf(stderr, "vgen_create failed: %d %s\n", rc, vgen_last_error(nullptr));
        exit(2);
    }
    return c;
}

static vgen_scan_config range_cfg(uint32_t fmt, uint64_t lo, uint64_t hi, uint64_t count = UINT64_MAX) {
    vgen_scan_config c;
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.format = fmt;
    c.count = count;
    c.has_start = 1;
    key_of(lo, c.start);
    c.has_end = 1;
    key_of(hi, c.end);
    return c;
}

// the oracle's matches of [lo, hi], sorted by key
static Pairs oracle_range(int fmt, const char *pat, int ci, uint64_t lo, uint64_t hi) {
    uint8_t a[32], b[32];
    key_of(lo, a);
    key_of(hi, b);
    vo_scan_result r;
    int rc = vo_scan_range(fmt, pat, ci, a, b, (size_t)-1, 0, &r);
    Pairs out;
    CHECK(rc == 0, "vo_scan_range rc=%d", rc);
    for (size_t i = 0; i < r.n_matches; i++) out.emplace_back(r.matches[i].gen.address, r.matches[i].gen.wif);
    vo_scan_free(&r);
    return out;
}

static Pairs got_of(const vgen_scan_result &r) {
    Pairs out;
    for (uint64_t i = 0; i < r.n_matches; i++) out.emplace_back(r.matches[i].address, r.matches[i].wif);
    return out;
}

struct CbLog {
    std::vector<uint64_t> seen;   // written by the scanning thread(s) under the library's lock / single thread
};
static void cb_log(uint64_t ops, void *u) { static_cast<CbLog *>(u)->seen.push_back(ops); }

// ---- scenarios ---------------------------------------------------------------------------------------------------

static void sc_range_scan() {
    const uint64_t lo = 0x30000, hi = lo + 10ull * BATCH - 1 - 77;   // the last batch is cut by `end`
    auto want = oracle_range(0, "^1[A-C]", 0, lo, hi);
    vgen_ctx *c = make_ctx(VGEN_FMT_P2PKH, 3);
    vgen_scan_config cfg = range_cfg(0, lo, hi);
    vgen_scan_result r;
    CbLog log;
    int rc = vgen_scan(c, "^1[A-C]", &cfg, cb_log, &log, nullptr, &r);
    CHECK(rc == VGEN_OK, "rc=%d %s", rc, vgen_last_error(c));
    CHECK(got_of(r) == want && want.size() > 100, "matches %zu vs %zu", (size_t)r.n_matches, want.size());
    CHECK(r.complete == 1 && r.operations == 10ull * BATCH && r.failed_shards == 0, "ops %llu", (unsigned long long)r.operations);
    CHECK(log.seen.size() == 10, "callbacks %zu", log.seen.size());
    for (size_t i = 0; i < log.seen.size(); i++) CHECK(log.seen[i] == (i + 1) * (uint64_t)BATCH, "callback %zu = %llu", i, (unsigned long long)log.seen[i]);
    vgen_scan_result_free(&r);
    // count-limited: the first five matches, in key order
    cfg.count = 5;
    rc = vgen_scan(c, "^1[A-C]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.n_matches == 5, "rc=%d n=%llu", rc, (unsigned long long)r.n_matches);
    auto g = got_of(r);
    CHECK(Pairs(want.begin(), want.begin() + 5) == g, "first five");
    vgen_scan_result_free(&r);
    vgen_destroy(c);
}

static void sc_stop_flag() {
    vgen_ctx *c = make_ctx(VGEN_FMT_P2PKH, 4);
    vgen_scan_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.format = 0;
    cfg.count = 1;
    cfg.seed = 7;
    volatile int32_t stop = 0;
    std::thread stopper([&]() {
        std::this_thread::sleep_for(std::chrono::milliseconds(150));
        __atomic_store_n((int32_t *)&stop, 1, __ATOMIC_RELAXED);
    });
    vgen_scan_result r;
    CbLog log;
    int rc = vgen_scan(c, "^1ZZZZZZZZZZ", &cfg, cb_log, &log, &stop, &r);   // scanner.rs:413-440
    stopper.join();
    CHECK(rc == VGEN_OK && r.n_matches == 0 && r.operations > 0 && r.operations % BATCH == 0, "rc=%d ops=%llu", rc, (unsigned long long)r.operations);
    CHECK(!log.seen.empty() && log.seen.back() == r.operations, "last callback");
    vgen_scan_result_free(&r);
    vgen_destroy(c);
}

static void sc_checkpoint() {
    const uint64_t lo = 0x50000, hi = lo + 9ull * BATCH - 1;
    auto want = oracle_range(0, "^1[D-F]", 0, lo, hi);
    char path[] = "/tmp/vgen_fake_ck_XXXXXX";
    int fd = mkstemp(path);
    close(fd);
    unlink(path);
    vgen_ctx *c = make_ctx(VGEN_FMT_P2PKH, 3);
    vgen_scan_config cfg = range_cfg(0, lo, hi);
    cfg.checkpoint_path = path;
    cfg.checkpoint_interval_ms = 1;
    cfg.max_batches = 4;
    vgen_scan_result r;
    int rc = vgen_scan(c, "^1[D-F]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.complete == 0 && r.operations == 4ull * BATCH, "first leg rc=%d ops=%llu", rc, (unsigned long long)r.operations);
    size_t first = (size_t)r.n_matches;
    vgen_scan_result_free(&r);
    cfg.max_batches = 0;
    rc = vgen_scan(c, "^1[D-F]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.complete == 1 && r.resumed_operations == 4ull * BATCH && r.operations == 5ull * BATCH, "resume rc=%d", rc);
    CHECK(got_of(r) == want && first > 0 && first < want.size(), "resumed matches %llu vs %zu", (unsigned long long)r.n_matches, want.size());
    vgen_scan_result_free(&r);
    // a finished checkpoint answers without touching the device
    rc = vgen_scan(c, "^1[D-F]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.operations == 0 && got_of(r) == want, "finished checkpoint");
    vgen_scan_result_free(&r);
    unlink(path);
    vgen_destroy(c);
}

static void sc_multi_context() {
    const uint64_t lo = 0x70000, hi = lo + 15ull * BATCH - 1;
    auto want = oracle_range(0, "^1[G-J]", 0, lo, hi);
    vgen_ctx *cs[3] = {make_ctx(0, 2), make_ctx(0, 3), make_ctx(0, 2)};
    vgen_scan_config cfg = range_cfg(0, lo, hi);
    vgen_scan_result r;
    CbLog log;
    int rc = vgen_scan_multi(cs, 3, "^1[G-J]", &cfg, cb_log, &log, nullptr, &r);
    CHECK(rc == VGEN_OK && got_of(r) == want && r.complete == 1 && r.operations == 15ull * BATCH, "multi rc=%d n=%llu/%zu", rc,
          (unsigned long long)r.n_matches, want.size());
    CHECK(log.seen.size() == 15, "callbacks %zu", log.seen.size());
    for (size_t i = 0; i < log.seen.size(); i++) CHECK(log.seen[i] == (i + 1) * (uint64_t)BATCH, "callback %zu", i);
    vgen_scan_result_free(&r);
    // with a checkpoint shared by the three shards
    char path[] = "/tmp/vgen_fake_mck_XXXXXX";
    int fd = mkstemp(path);
    close(fd);
    unlink(path);
    cfg.checkpoint_path = path;
    cfg.checkpoint_interval_ms = 1;
    cfg.max_batches = 2;
    rc = vgen_scan_multi(cs, 3, "^1[G-J]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.complete == 0 && r.operations == 6ull * BATCH, "multi ck leg 1 rc=%d", rc);
    vgen_scan_result_free(&r);
    cfg.max_batches = 0;
    rc = vgen_scan_multi(cs, 3, "^1[G-J]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.complete == 1 && got_of(r) == want, "multi ck leg 2 rc=%d", rc);
    vgen_scan_result_free(&r);
    unlink(path);
    for (auto *c : cs) vgen_destroy(c);
}

static void sc_ring_growth_and_host_filter() {
    const uint64_t lo = 1, hi = 4ull * BATCH;
    vgen_ctx *c = make_ctx(0, 3, 256);
    const char *pats[] = {"^1[A-F]", "^1[2-9A-Za-z]", "1[A-D][a-z]"};   // ring grows / dumps + host pool / on-device DFA with overflow
    for (const char *p : pats) {
        auto want = oracle_range(0, p, 0, lo, hi);
        vgen_scan_config cfg = range_cfg(0, lo, hi);
        vgen_scan_result r;
        int rc = vgen_scan(c, p, &cfg, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_OK && got_of(r) == want && want.size() > 1000, "%s: rc=%d n=%llu/%zu", p, rc, (unsigned long long)r.n_matches, want.size());
        vgen_scan_result_free(&r);
    }
    vgen_destroy(c);
}

static void sc_failure_takeover() {
    const uint64_t lo = 0x90000, hi = lo + 18ull * BATCH - 1;
    auto want = oracle_range(0, "^1[K-N]", 0, lo, hi);
    vgen_scan_config cfg = range_cfg(0, lo, hi);
    for (int victim = 0; victim < 3; victim++) {
        vgen_ctx *cs[3] = {make_ctx(0, 2), make_ctx(0, 2), make_ctx(0, 3)};
        vgen_debug_fail_after(cs[victim], (uint64_t)victim * 2);
        vgen_scan_result r;
        int rc = vgen_scan_multi(cs, 3, "^1[K-N]", &cfg, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_OK && got_of(r) == want && r.complete == 1 && r.failed_shards == 1, "victim %d: rc=%d n=%llu/%zu failed=%d", victim, rc,
              (unsigned long long)r.n_matches, want.size(), r.failed_shards);
        vgen_scan_result_free(&r);
        for (auto *c : cs) vgen_destroy(c);
    }
    vgen_ctx *cs[3] = {make_ctx(0, 2), make_ctx(0, 2), make_ctx(0, 2)};
    vgen_debug_fail_after(cs[0], 1);
    vgen_debug_fail_after(cs[1], 4);
    vgen_debug_fail_after(cs[2], 2);
    vgen_scan_result r;
    int rc = vgen_scan_multi(cs, 3, "^1[K-N]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_E_HIP && r.complete == 0 && r.failed_shards == 3, "all fail: rc=%d failed=%d", rc, r.failed_shards);
    CHECK(strstr(vgen_last_error(cs[0]), "injected device failure") != nullptr, "message: %s", vgen_last_error(cs[0]));
    auto g = got_of(r);
    CHECK(!g.empty() && g.size() < want.size(), "partial %zu of %zu", g.size(), want.size());
    for (auto &m : g) CHECK(std::find(want.begin(), want.end(), m) != want.end(), "partial match not in the oracle's set: %s", m.first.c_str());
    vgen_scan_result_free(&r);
    // single context: error + the finished batches' matches, a prefix of the oracle's list
    for (auto *c : cs) vgen_debug_fail_after(c, UINT64_MAX);
    vgen_debug_fail_after(cs[1], 5);
    rc = vgen_scan(cs[1], "^1[K-N]", &cfg, nullptr, nullptr, nullptr, &r);
    g = got_of(r);
    CHECK(rc == VGEN_E_HIP && r.failed_shards == 1 && !g.empty() && g.size() < want.size(), "single: rc=%d n=%zu", rc, g.size());
    CHECK(Pairs(want.begin(), want.begin() + g.size()) == g, "single: prefix");
    vgen_scan_result_free(&r);
    for (auto *c : cs) vgen_destroy(c);
}

static void sc_random_keys() {
    vgen_ctx *c = make_ctx(0, 2);
    vgen_scan_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.count = 4;
    cfg.seed = 42;
    cfg.flags = VGEN_SCAN_RANDOM_KEYS;
    vgen_scan_result r;
    int rc = vgen_scan(c, "^1A", &cfg, nullptr, nullptr, nullptr, &r);
    vo_scan_result o;
    vo_scan_random(0, "^1A", 0, 42, 4, 0, 1, &o);
    CHECK(rc == VGEN_OK && r.n_matches == 4 && o.n_matches == 4, "random rc=%d", rc);
    for (size_t i = 0; i < 4 && i < r.n_matches && i < o.n_matches; i++)
        CHECK(!strcmp(r.matches[i].address, o.matches[i].gen.address) && !memcmp(r.matches[i].key, o.matches[i].key, 32), "random match %zu", i);
    vo_scan_free(&o);
    vgen_scan_result_free(&r);
    // a pattern nearly every candidate matches: full dumps filtered on the host pool — the oracle's first fifty, in order
    cfg.count = 50;
    rc = vgen_scan(c, "^1[1-9A-Za-z]", &cfg, nullptr, nullptr, nullptr, &r);
    vo_scan_random(0, "^1[1-9A-Za-z]", 0, 42, 50, 0, 1, &o);
    CHECK(rc == VGEN_OK && r.n_matches == 50 && o.n_matches == 50, "random host-filter rc=%d n=%llu", rc, (unsigned long long)r.n_matches);
    for (size_t i = 0; i < 50 && i < r.n_matches && i < o.n_matches; i++)
        CHECK(!strcmp(r.matches[i].address, o.matches[i].gen.address) && !memcmp(r.matches[i].key, o.matches[i].key, 32), "random host-filter match %zu", i);
    vo_scan_free(&o);
    vgen_scan_result_free(&r);
    vgen_destroy(c);
    // three striped contexts, shard i walking stream i; and endomorphism contexts (six keys per draw): every match re-derives
    for (uint32_t flags : {0u, (uint32_t)VGEN_FLAG_ENDO}) {
        vgen_ctx *cs[3] = {make_ctx(0, 2, 0, flags), make_ctx(0, 2, 0, flags), make_ctx(0, 2, 0, flags)};
        cfg.count = 9;
        rc = vgen_scan_multi(cs, 3, "^1A", &cfg, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_OK && r.n_matches == 9, "multi random (flags %u) rc=%d n=%llu", flags, rc, (unsigned long long)r.n_matches);
        for (uint64_t i = 0; i < r.n_matches; i++) {
            vo_generated g;
            CHECK(vo_generate(0, r.matches[i].key, &g) && !strcmp(g.address, r.matches[i].address), "multi random match %llu", (unsigned long long)i);
            for (uint64_t j = 0; j < i; j++) CHECK(memcmp(r.matches[i].key, r.matches[j].key, 32) != 0, "duplicate key %llu/%llu", (unsigned long long)i, (unsigned long long)j);
        }
        vgen_scan_result_free(&r);
        for (auto *x : cs) vgen_destroy(x);
    }
}

// A random-key scan with a checkpoint: interrupted after three batches, resumed, it returns what the oracle's walk of the same
// stream returns — seeded, and unseeded (the file then carries the seed the first leg drew); a key-range checkpoint is refused.
static void sc_random_checkpoint() {
    for (uint64_t seed : {77ull, 0ull}) {
        char path[] = "/tmp/vgen_fake_rck_XXXXXX";
        int fd = mkstemp(path);
        close(fd);
        unlink(path);
        vgen_ctx *c = make_ctx(0, 3);
        vgen_scan_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.count = UINT64_MAX;
        cfg.seed = seed;
        cfg.flags = VGEN_SCAN_RANDOM_KEYS;
        cfg.checkpoint_path = path;
        cfg.checkpoint_interval_ms = 1;
        cfg.max_batches = 1;
        vgen_scan_result r;
        int rc = vgen_scan(c, "^1[A-D][a-k]", &cfg, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_OK && r.operations == 1ull * BATCH && r.complete == 0, "leg 1 (seed %llu): rc=%d ops=%llu %s", (unsigned long long)seed, rc,
              (unsigned long long)r.operations, vgen_last_error(c));
        const uint64_t first = r.n_matches;
        vgen_scan_result_free(&r);
        cfg.max_batches = 2;   // (per call: two more batches, three in all)
        rc = vgen_scan(c, "^1[A-D][a-k]", &cfg, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_OK && r.resumed_operations == 1ull * BATCH && r.operations == 2ull * BATCH, "leg 2 (seed %llu): rc=%d resumed=%llu ops=%llu %s",
              (unsigned long long)seed, rc, (unsigned long long)r.resumed_operations, (unsigned long long)r.operations, vgen_last_error(c));
        CHECK(first > 0 && r.n_matches > first, "matches %llu then %llu", (unsigned long long)first, (unsigned long long)r.n_matches);
        if (seed) {
            // the oracle walks its stream in batches of 10 000 keys (scanner.rs:107): 3 x 8192 = 24 576 keys lie between its 20 000
            // and its 30 000 — the product's matches must extend the former's and be a prefix of the latter's
            vo_scan_result lo, hi;
            vo_scan_random(0, "^1[A-D][a-k]", 0, seed, (size_t)-1, 20000, 1, &lo);
            vo_scan_random(0, "^1[A-D][a-k]", 0, seed, (size_t)-1, 3ull * BATCH, 1, &hi);
            CHECK(lo.n_matches <= r.n_matches && r.n_matches <= hi.n_matches && hi.n_matches > lo.n_matches && lo.n_matches > 100,
                  "resumed random scan: %llu matches, the oracle's %zu (20 000 keys) .. %zu (30 000)", (unsigned long long)r.n_matches, lo.n_matches, hi.n_matches);
            for (size_t i = 0; i < r.n_matches && i < hi.n_matches; i++)
                CHECK(!memcmp(r.matches[i].key, hi.matches[i].key, 32) && !strcmp(r.matches[i].address, hi.matches[i].gen.address), "resumed random match %zu", i);
            vo_scan_free(&lo);
            vo_scan_free(&hi);
        } else {
            for (uint64_t i = 0; i < r.n_matches; i++) {   // the drawn seed is the file's business: every match must re-derive, none twice
                vo_generated g;
                CHECK(vo_generate(0, r.matches[i].key, &g) && !strcmp(g.address, r.matches[i].address), "unseeded resumed match %llu", (unsigned long long)i);
                for (uint64_t j = 0; j < i; j++) CHECK(memcmp(r.matches[i].key, r.matches[j].key, 32) != 0, "duplicate %llu/%llu", (unsigned long long)i, (unsigned long long)j);
            }
        }
        vgen_scan_result_free(&r);
        // the same file under another seed, or as a key-range scan: refused
        vgen_scan_config other = cfg;
        other.seed = seed + 5;
        rc = vgen_scan(c, "^1[A-D][a-k]", &other, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_E_INVALID, "another seed: rc=%d", rc);
        vgen_scan_result_free(&r);
        other = range_cfg(0, 1, 100);
        other.checkpoint_path = path;
        rc = vgen_scan(c, "^1[A-D][a-k]", &other, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_E_INVALID, "as a range scan: rc=%d", rc);
        vgen_scan_result_free(&r);
        unlink(path);
        vgen_destroy(c);
    }
}

static void sc_endo_and_formats() {
    // six images per point: every match must re-derive on the oracle from its reported key
    vgen_ctx *c = make_ctx(0, 3, 0, VGEN_FLAG_ENDO);
    vgen_scan_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.count = 12;
    vgen_scan_result r;
    int rc = vgen_scan(c, "^1B", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.n_matches == 12 && r.operations % (6ull * BATCH) == 0, "endo rc=%d", rc);
    for (uint64_t i = 0; i < r.n_matches; i++) {
        vo_generated g;
        CHECK(vo_generate(0, r.matches[i].key, &g) && !strcmp(g.address, r.matches[i].address) && !strcmp(g.wif, r.matches[i].wif), "endo match %llu",
              (unsigned long long)i);
    }
    vgen_scan_result_free(&r);
    vgen_destroy(c);
    // Bech32 suffix (checksum tables), Ethereum case-insensitive, P2SH: range scans against the oracle
    struct { uint32_t fmt; const char *pat; int ci; } cases[] = {{1, "aa$", 0}, {5, "^0xab", 1}, {2, "^3[A-D]", 0}, {4, "^1[P-R]", 0}};
    for (auto &k : cases) {
        const uint64_t lo = 0xB0000 + k.fmt, hi = lo + 5ull * BATCH - 1;
        auto want = oracle_range((int)k.fmt, k.pat, k.ci, lo, hi);
        vgen_ctx *cc = make_ctx(k.fmt, 2);
        vgen_scan_config cf = range_cfg(k.fmt, lo, hi);
        cf.case_insensitive = k.ci;
        int rc2 = vgen_scan(cc, k.pat, &cf, nullptr, nullptr, nullptr, &r);
        CHECK(rc2 == VGEN_OK && got_of(r) == want && !want.empty(), "fmt %u %s: rc=%d n=%llu/%zu", k.fmt, k.pat, rc2, (unsigned long long)r.n_matches, want.size());
        vgen_scan_result_free(&r);
        vgen_destroy(cc);
    }
}

static void sc_dispatch_api() {
    // the frame-level API: dump mode, explicit keys, state errors
    vgen_ctx *c = make_ctx(0, 2);
    uint8_t k0[32];
    key_of(12345, k0);
    CHECK(vgen_set_filter(c, nullptr) == VGEN_OK, "dump mode");
    CHECK(vgen_dispatch(c, 0, k0) == VGEN_OK && vgen_dispatch(c, 0, k0) == VGEN_E_STATE, "double dispatch");
    uint64_t tested = 0;
    CHECK(vgen_wait(c, 0, nullptr, 0, nullptr, &tested) == VGEN_OK && tested == BATCH, "wait");
    CHECK(vgen_wait(c, 0, nullptr, 0, nullptr, nullptr) == VGEN_E_STATE, "wait twice");
    std::vector<uint8_t> dump((size_t)BATCH * 20), ref((size_t)BATCH * 20);
    CHECK(vgen_read_dump(c, 0, dump.data(), dump.size()) == VGEN_OK, "read dump");
    vo_payload_seq(0, k0, BATCH, 0, ref.data());
    CHECK(dump == ref, "dump parity");
    std::vector<uint8_t> keys(64 * 32);
    for (int i = 0; i < 64; i++) key_of(0x1000 + 977ull * i * i, keys.data() + 32 * i);
    memset(keys.data() + 32 * 5, 0, 32);   // an invalid scalar yields nothing
    CHECK(vgen_dispatch_keys(c, 1, keys.data(), 64) == VGEN_OK && vgen_wait(c, 1, nullptr, 0, nullptr, &tested) == VGEN_OK && tested == 64, "keys");
    CHECK(vgen_read_dump(c, 1, dump.data(), dump.size()) == VGEN_OK, "read dump 2");
    for (int i = 0; i < 64; i++) {
        uint8_t pl[32] = {0};
        int n = i == 5 ? 0 : vo_payload(0, keys.data() + 32 * i, pl);
        CHECK((n == 20 || i == 5) && !memcmp(dump.data() + 20 * i, pl, 20), "key %d", i);
    }
    uint8_t bad[32] = {0};
    CHECK(vgen_dispatch(c, 0, bad) == VGEN_E_RANGE && vgen_dispatch(c, 9, k0) == VGEN_E_INVALID, "bad arguments");
    vgen_destroy(c);
}

// 32-byte big-endian from a hex string
static void key_hex(const char *hex, uint8_t be[32]) {
    memset(be, 0, 32);
    const size_t n = strlen(hex);
    for (size_t i = 0; i < n; i++) {
        const char ch = hex[n - 1 - i];
        const uint8_t v = (uint8_t)(ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10);
        be[31 - i / 2] |= (uint8_t)(v << (4 * (i % 2)));
    }
}

static void sc_edge_ranges() {
    // ranges that are not whole batches: shorter than one, a single key, ending at the last valid scalar n - 1 (the batches
    // that touch the group order take the per-key path and the key space runs out), end below start
    vgen_ctx *c = make_ctx(0, 3);
    struct { const char *lo, *hi; const char *pat; } cases[] = {
        {"5", "64", "^1"},
        {"1", "1", "^1"},
        {"ffff", "ffff", "^1[A-Z]"},
        {"fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0360000", "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364140", "^1[A-F]"},
        {"fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364140", "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364140", "^1"},
    };
    for (auto &k : cases) {
        vgen_scan_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.count = UINT64_MAX;
        cfg.has_start = cfg.has_end = 1;
        key_hex(k.lo, cfg.start);
        key_hex(k.hi, cfg.end);
        vo_scan_result o;
        const int orc = vo_scan_range(0, k.pat, 0, cfg.start, cfg.end, (size_t)-1, 0, &o);
        vgen_scan_result r;
        const int rc = vgen_scan(c, k.pat, &cfg, nullptr, nullptr, nullptr, &r);
        CHECK(rc == VGEN_OK && orc == 0 && r.n_matches == o.n_matches && r.complete == 1, "%s..%s: rc=%d n=%llu/%zu complete=%d", k.lo, k.hi, rc,
              (unsigned long long)r.n_matches, o.n_matches, r.complete);
        for (size_t i = 0; i < r.n_matches && i < o.n_matches; i++)
            CHECK(!memcmp(r.matches[i].key, o.matches[i].key, 32) && !strcmp(r.matches[i].wif, o.matches[i].gen.wif), "%s..%s match %zu", k.lo, k.hi, i);
        vo_scan_free(&o);
        vgen_scan_result_free(&r);
    }
    // end below start: nothing to scan, no error, range "complete"
    vgen_scan_config cfg = range_cfg(0, 1000, 10);
    vgen_scan_result r;
    int rc = vgen_scan(c, "^1", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && r.n_matches == 0 && r.operations == 0, "end < start: rc=%d ops=%llu", rc, (unsigned long long)r.operations);
    vgen_scan_result_free(&r);
    // start = 0 or >= n is refused (SecretKey::from_slice, gpu.rs:903)
    cfg = range_cfg(0, 0, 10);
    rc = vgen_scan(c, "^1", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_E_RANGE, "start 0: rc=%d", rc);
    vgen_scan_result_free(&r);
    // the same short ranges striped over three contexts (most shards have nothing to do)
    vgen_ctx *cs[3] = {c, make_ctx(0, 2), make_ctx(0, 2)};
    cfg = range_cfg(0, 5, 5 + 2ull * BATCH + 17);
    auto want = oracle_range(0, "^1[A-H]", 0, 5, 5 + 2ull * BATCH + 17);
    rc = vgen_scan_multi(cs, 3, "^1[A-H]", &cfg, nullptr, nullptr, nullptr, &r);
    CHECK(rc == VGEN_OK && got_of(r) == want && r.complete == 1, "short striped range: rc=%d n=%llu/%zu", rc, (unsigned long long)r.n_matches, want.size());
    vgen_scan_result_free(&r);
    for (auto *x : cs) vgen_destroy(x);
}

// Randomised differential: formats x pattern kinds (hash160 ranges, bit masks, on-device automaton, full dumps filtered on the
// host, NFA-walk patterns) x ranges cut anywhere x counts x one to three contexts x frames x ring sizes x injected failures,
// every result against the oracle's scan of the same range.  VGEN_FAKE_FUZZ_SEED / VGEN_FAKE_FUZZ_CASES select the walk.
static void sc_fuzz() {
    struct Pat { uint32_t fmt; const char *p; int ci; };
    static const Pat pats[] = {
        {0, "^1[A-C]", 0}, {0, "^1", 0}, {0, "[A-Z]{2}", 0}, {0, "Q$", 0}, {0, "a.{20}$", 0}, {0, "^1c", 1}, {0, "^1[Oo]", 0},
        {1, "^bc1q[ac]", 0}, {1, "a$", 0}, {1, "xy", 0}, {1, "^BC1Q[AC]", 1},
        {2, "^3[A-C]", 0}, {2, "^3", 0}, {2, "z$", 0},
        {4, "^1[D-F]", 0}, {4, "[a-f]{3}", 0},
        {5, "^0x[0-3]", 0}, {5, "^0xA", 1}, {5, "f$", 0}, {5, "^0x[A-F]", 0},
        {3, "^bc1p[ac]", 0}, {3, "q$", 0},
    };
    const char *es = getenv("VGEN_FAKE_FUZZ_SEED"), *ec = getenv("VGEN_FAKE_FUZZ_CASES");
    uint64_t x = es ? strtoull(es, nullptr, 0) : 20261004ull;
    const int cases = ec ? atoi(ec) : 12;
    auto rnd = [&x](uint64_t n) {   // xorshift64*
        x ^= x >> 12;
        x ^= x << 25;
        x ^= x >> 27;
        return (x * 2685821657736338717ull >> 16) % n;
    };
    for (int t = 0; t < cases; t++) {
        const Pat &pt = pats[rnd(sizeof pats / sizeof pats[0])];
        const uint64_t lo = 1 + rnd(2) * rnd(1ull << 40) + rnd(3 * BATCH);
        const uint64_t len = 1 + rnd(4 * BATCH) + rnd(2) * rnd(BATCH);
        const uint64_t hi = lo + len - 1;
        static const uint64_t counts[] = {1, 3, 50, UINT64_MAX, UINT64_MAX};
        const uint64_t count = counts[rnd(5)];
        const uint32_t n_ctx = 1 + (uint32_t)rnd(3);
        const uint32_t cap = rnd(2) ? 0 : 256;
        vgen_ctx *cs[3] = {nullptr, nullptr, nullptr};
        for (uint32_t i = 0; i < n_ctx; i++) cs[i] = make_ctx(pt.fmt, 2 + (uint32_t)rnd(4), cap);
        int victim = -1;
        if (n_ctx > 1 && rnd(3) == 0) {
            victim = (int)rnd(n_ctx);
            vgen_debug_fail_after(cs[victim], rnd(3));
        }
        if (pt.fmt != 3 && rnd(8) == 0) {
            // a vanity search on endomorphism contexts (six images per point, random bases): every match re-derives on the
            // oracle from its key and satisfies the pattern, no key twice
            for (uint32_t i = 0; i < n_ctx; i++) vgen_destroy(cs[i]);
            for (uint32_t i = 0; i < n_ctx; i++) cs[i] = make_ctx(pt.fmt, 2 + (uint32_t)rnd(3), cap, VGEN_FLAG_ENDO);
            vgen_scan_config e;
            memset(&e, 0, sizeof e);
            e.struct_size = sizeof e;
            e.format = pt.fmt;
            e.case_insensitive = pt.ci;
            e.count = 1 + rnd(20);
            e.max_batches = 40;
            vgen_scan_result r;
            const int rc = n_ctx == 1 ? vgen_scan(cs[0], pt.p, &e, nullptr, nullptr, nullptr, &r) : vgen_scan_multi(cs, n_ctx, pt.p, &e, nullptr, nullptr, nullptr, &r);
            CHECK(rc == VGEN_OK && r.n_matches <= e.count, "case %d: endo fmt %u '%s' ctx %u: rc=%d n=%llu", t, pt.fmt, pt.p, n_ctx, rc, (unsigned long long)r.n_matches);
            char rerr[128];
            vo_regex *re = vo_regex_new(pt.p, pt.ci, rerr, sizeof rerr);
            for (uint64_t i = 0; i < r.n_matches; i++) {
                vo_generated g;
                CHECK(vo_generate((int)pt.fmt, r.matches[i].key, &g) && !strcmp(g.address, r.matches[i].address) && !strcmp(g.wif, r.matches[i].wif),
                      "case %d: endo match %llu does not re-derive", t, (unsigned long long)i);
                CHECK(re && vo_regex_is_match(re, r.matches[i].address), "case %d: endo match %s does not satisfy '%s'", t, r.matches[i].address, pt.p);
                for (uint64_t j = 0; j < i; j++) CHECK(memcmp(r.matches[i].key, r.matches[j].key, 32) != 0, "case %d: endo duplicate key", t);
            }
            if (re) vo_regex_free(re);
            vgen_scan_result_free(&r);
            for (uint32_t i = 0; i < n_ctx; i++) vgen_destroy(cs[i]);
            continue;
        }
        if (n_ctx == 1 && rnd(5) == 0) {
            // an independent random key per candidate (seeded stream): the oracle's walk of the same stream, in order
            vgen_scan_config rc_cfg;
            memset(&rc_cfg, 0, sizeof rc_cfg);
            rc_cfg.struct_size = sizeof rc_cfg;
            rc_cfg.format = pt.fmt;
            rc_cfg.case_insensitive = pt.ci;
            rc_cfg.count = 1 + rnd(6);
            rc_cfg.seed = 1 + rnd(1000);
            rc_cfg.flags = VGEN_SCAN_RANDOM_KEYS;
            rc_cfg.max_batches = 6;
            vgen_scan_result r;
            const int rc = vgen_scan(cs[0], pt.p, &rc_cfg, nullptr, nullptr, nullptr, &r);
            vo_scan_result o;
            const int orc = vo_scan_random((int)pt.fmt, pt.p, pt.ci, rc_cfg.seed, (size_t)rc_cfg.count, 6ull * BATCH, 1, &o);
            CHECK(rc == VGEN_OK && orc == 0 && r.n_matches == o.n_matches, "case %d: random keys fmt %u '%s' seed %llu count %llu: rc=%d n=%llu/%zu", t, pt.fmt, pt.p,
                  (unsigned long long)rc_cfg.seed, (unsigned long long)rc_cfg.count, rc, (unsigned long long)r.n_matches, o.n_matches);
            for (size_t i = 0; i < r.n_matches && i < o.n_matches; i++)
                CHECK(!strcmp(r.matches[i].address, o.matches[i].gen.address) && !memcmp(r.matches[i].key, o.matches[i].key, 32), "case %d: random match %zu", t, i);
            vo_scan_free(&o);
            vgen_scan_result_free(&r);
            vgen_destroy(cs[0]);
            continue;
        }
        auto want = oracle_range((int)pt.fmt, pt.p, pt.ci, lo, hi);
        vgen_scan_config cfg = range_cfg(pt.fmt, lo, hi, count);
        cfg.case_insensitive = pt.ci;
        vgen_scan_result r;
        // every fourth unbounded case through a checkpoint: a first leg of a few batches per context, then the rest
        char ckpath[] = "/tmp/vgen_fake_fz_XXXXXX";
        const bool with_ck = count == UINT64_MAX && rnd(4) == 0;
        if (with_ck) {
            int fd = mkstemp(ckpath);
            close(fd);
            unlink(ckpath);
            cfg.checkpoint_path = ckpath;
            cfg.checkpoint_interval_ms = 1;
            cfg.max_batches = 1 + rnd(2);
            const int rc1 = n_ctx == 1 ? vgen_scan(cs[0], pt.p, &cfg, nullptr, nullptr, nullptr, &r)
                                       : vgen_scan_multi(cs, n_ctx, pt.p, &cfg, nullptr, nullptr, nullptr, &r);
            CHECK(rc1 == VGEN_OK || victim >= 0, "case %d: checkpoint leg 1 rc=%d %s", t, rc1, vgen_last_error(cs[0]));
            vgen_scan_result_free(&r);
            cfg.max_batches = 0;
            if (victim >= 0) vgen_debug_fail_after(cs[victim], UINT64_MAX);   // (the failure belongs to the first leg)
        }
        const int rc = n_ctx == 1 ? vgen_scan(cs[0], pt.p, &cfg, nullptr, nullptr, nullptr, &r)
                                  : vgen_scan_multi(cs, n_ctx, pt.p, &cfg, nullptr, nullptr, nullptr, &r);
        if (with_ck) unlink(ckpath);
        auto got = got_of(r);
        const size_t expect = (size_t)std::min<uint64_t>(count, want.size());
        char what[256];
        snprintf(what, sizeof what, "case %d: fmt %u '%s'%s [%llx, +%llu] count %lld ctx %u cap %u victim %d: rc=%d n=%zu/%zu (want %zu)", t, pt.fmt, pt.p,
                 pt.ci ? " -i" : "", (unsigned long long)lo, (unsigned long long)len, (long long)count, n_ctx, cap, victim, rc, got.size(), expect, want.size());
        CHECK(rc == VGEN_OK, "%s: %s", what, vgen_last_error(cs[0]));
        CHECK(got.size() == expect, "%s", what);
        if (n_ctx == 1 || count == UINT64_MAX) {
            // one context walks in key order; an unbounded scan of several returns everything, sorted
            CHECK(got == Pairs(want.begin(), want.begin() + std::min(expect, want.size())), "%s: not the oracle's first matches", what);
        } else {
            // several contexts racing to `count`: which of the range's matches made it depends on their speeds
            for (auto &m : got) CHECK(std::find(want.begin(), want.end(), m) != want.end(), "%s: %s is not a match of the range", what, m.first.c_str());
            for (size_t i = 1; i < got.size(); i++) CHECK(got[i] != got[i - 1], "%s: duplicate", what);
        }
        if (count == UINT64_MAX) {
            const uint64_t batches = (len + BATCH - 1) / BATCH;
            CHECK(r.complete == 1 && r.operations + r.resumed_operations == batches * BATCH && (with_ck || r.resumed_operations == 0), "%s: complete=%d ops=%llu+%llu%s", what,
                  r.complete, (unsigned long long)r.resumed_operations, (unsigned long long)r.operations, with_ck ? " (checkpoint)" : "");
        }
        // (the victim fails only if it gets to its k-th dispatch before the scan ends)
        CHECK(r.failed_shards == 0 || (victim >= 0 && !with_ck && r.failed_shards == 1), "%s: failed_shards=%d", what, r.failed_shards);
        vgen_scan_result_free(&r);
        for (uint32_t i = 0; i < n_ctx; i++) vgen_destroy(cs[i]);
    }
}

int main(int argc, char **argv) {
    const std::map<std::string, std::function<void()>> all = {
        {"range_scan", sc_range_scan}, {"stop_flag", sc_stop_flag}, {"checkpoint", sc_checkpoint}, {"multi_context", sc_multi_context},
        {"ring_growth", sc_ring_growth_and_host_filter}, {"failure_takeover", sc_failure_takeover}, {"random_keys", sc_random_keys},
        {"endo_and_formats", sc_endo_and_formats}, {"dispatch_api", sc_dispatch_api}, {"edge_ranges", sc_edge_ranges}, {"fuzz", sc_fuzz}, {"random_checkpoint", sc_random_checkpoint}};
    std::vector<std::string> run;
    for (int i = 1; i < argc; i++) run.push_back(argv[i]);
    if (run.empty())
        for (auto &kv : all) run.push_back(kv.first);
    for (auto &name : run) {
        auto it = all.find(name);
        if (it == all.end()) {
            fprintf(stderr, "unknown scenario %s\n", name.c_str());
            return 2;
        }
        const int before = g_fail;
        const auto t0 = std::chrono::steady_clock::now();
        it->second();
        printf("%-18s %s  (%.1f s)\n", name.c_str(), g_fail == before ? "ok" : "FAILED",
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        fflush(stdout);
    }
    return g_fail ? 1 : 0;
}
