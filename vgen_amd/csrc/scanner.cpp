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
#include <fcntl.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <memory>
#include <random>
#include <thread>
#include <vector>

#include "../../include/vgen_hip.h"
#include "host/encode.h"
#include "host/filter.h"
#include "host/scalar.h"
#include "runtime.h"

// -DVGEN_SCAN_PROFILE (tools/permissive_probe.py, profiles/r04_permissive.txt): seconds the scanning thread spends in vgen_wait, in the
// worker pool's confirmation pass, in the hand-over of the workers' matches and in dispatch calls, printed when a shard ends.
#ifdef VGEN_SCAN_PROFILE
static inline double prof_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define PROF(var, stmt) { const double t_ = prof_now(); stmt; var += prof_now() - t_; }
#else
#define PROF(var, stmt) { stmt; }
#endif

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


void random_valid_key(Scalar &k) {
    std::random_device rd;
    for (;;) {
        for (int i = 0; i < 8; i++) k.w[i] = rd();
        if (scalar_is_valid(k)) return;   // rejection sampling, gpu.rs:938-944
    }
}

// `images`: 1, or 6 when the dispatch tested the endomorphism / negation images of every point: index is then
// variant * batch + i and the key is variant `index / batch` of batch_start + i (host/scalar.h).
// How a batch's candidate index maps to its private key: batch_start + index (the walk), or the counter-based stream.
struct BatchKeys {
    Scalar start{};
    bool random = false;
    RndSeed seed{};
    uint64_t first_index = 0;
    uint32_t stream = 0;
};

// The seed of a random-key scan (core/rnd.h): the caller's 64-bit one — reproducible runs and tests, NOT for keys that will
// hold value — or, unseeded, 192 bits of OS entropy (every key the mode returns is a function of seed, stream and index:
// the seed is all the secret there is; the reference seeds a 256-bit StdRng from the OS, src/scanner.rs:144).
RndSeed scan_rnd_seed(uint64_t cfg_seed) {
    if (cfg_seed) return rnd_seed_from_u64(cfg_seed);
    std::random_device rd;   // (getrandom / /dev/urandom on this platform)
    RndSeed s;
    for (int i = 0; i < 6; i++) s.w[i] = rd();
    return s;
}

// A confirmed match as the scan loop carries it: the private key and the address string.  WIF and hex are rendered when
// the match is handed to the caller (finish_result), in parallel and only for the matches that survive `count` — the
// reference, too, builds its WIF per MATCH, not per candidate (src/gpu.rs:1080-1088).  [Round 3 carried the 276-byte ABI
// record with everything rendered: on a permissive pattern (`^1C`: one key in 23) more than half of a scan was the serial
// hand-over of those records and their second Base58Check, profiles/r04_permissive.txt.]
struct LiteMatch {
    uint8_t key[32];
    char address[64];     // NUL-terminated (longest: a 62-character bech32m address)
};

// The matches of a scan in hand-over order, kept as the BLOCKS they arrive in (the worker threads' per-batch results, moved in whole): a
// permissive pattern yields millions, and copying them into one growing vector — reallocations and first-touch page faults, all on the
// scanning thread — was a third of such a scan even after the records had shrunk to 96 bytes (profiles/r04_permissive.txt).
class MatchList {
public:
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void push_back(const LiteMatch &m) {
        if (blocks_.empty() || blocks_.back().size() == blocks_.back().capacity()) {
            blocks_.emplace_back();
            blocks_.back().reserve(1024);
        }
        blocks_.back().push_back(m);
        n_++;
    }
    // the first k entries of v, without copying them
    void take(std::vector<LiteMatch> &&v, size_t k) {
        if (k == 0) return;
        if (k < v.size()) v.resize(k);
        n_ += v.size();
        blocks_.push_back(std::move(v));
    }
    void append_copy(const std::vector<LiteMatch> &v) {
        if (v.empty()) return;
        n_ += v.size();
        blocks_.push_back(v);
    }
    void append(MatchList &&o) {
        for (auto &b : o.blocks_) {
            n_ += b.size();
            blocks_.push_back(std::move(b));
        }
        o.blocks_.clear();
        o.n_ = 0;
    }
    void truncate(size_t n) {   // keep the first n
        while (n_ > n) {
            auto &b = blocks_.back();
            const size_t drop = std::min(n_ - n, b.size());
            b.resize(b.size() - drop);
            n_ -= drop;
            if (b.empty()) blocks_.pop_back();
        }
    }
    std::vector<LiteMatch> flatten() const {
        std::vector<LiteMatch> out;
        out.reserve(n_);
        for (auto &b : blocks_) out.insert(out.end(), b.begin(), b.end());
        return out;
    }
    const std::vector<std::vector<LiteMatch>> &blocks() const { return blocks_; }

private:
    std::vector<std::vector<LiteMatch>> blocks_;
    size_t n_ = 0;
};

bool make_match(const vgen_filter &flt, uint32_t format, const BatchKeys &bk, uint32_t index,
                const uint8_t *payload, const Scalar *end, LiteMatch &g, uint32_t batch = 0, uint32_t images = 1) {
    std::string addr = address_from_payload(format, payload);
    if (addr.empty() || addr.size() >= sizeof g.address || !flt.dfa.is_match(addr)) return false;          // pattern.matches, gpu.rs:1069
    Scalar k;
    const uint32_t variant = images > 1 ? index / batch : 0;
    if (images > 1) index %= batch;
    if (bk.random) {
        uint8_t rk[32];
        if (!random_key_be(bk.seed, bk.stream, bk.first_index + index, rk)) return false;   // not a valid draw: no key
        scalar_from_be(k, rk);
    } else if (scalar_add_u64(k, bk.start, index) || !scalar_is_valid(k)) {
        return false;                                                   // increment_key -> None
    }
    if (variant) {
        Scalar kv;
        scalar_variant(kv, k, variant);
        k = kv;
    }
    if (end && scalar_cmp(k, *end) > 0) return false;                   // gpu.rs:1074-1078
    scalar_to_be(k, g.key);
    memcpy(g.address, addr.c_str(), addr.size() + 1);
    return true;
}

// GeneratedAddress (src/address.rs:63-72) of a match: address, WIF (src/gpu.rs:1080-1088), hex, format, key.
void render_match(uint32_t format, const LiteMatch &m, vgen_generated &g) {
    memset(&g, 0, sizeof g);
    const std::string wif = key_to_wif(format, m.key), hex = hex_lower(m.key, 32);
    strncpy(g.address, m.address, sizeof g.address - 1);
    strncpy(g.wif, wif.c_str(), sizeof g.wif - 1);
    strncpy(g.hex, hex.c_str(), sizeof g.hex - 1);
    g.format = format;
    memcpy(g.key, m.key, 32);
}

// a recorded match (checkpoint file: keys only) back into the loop's form
bool lite_from_key(uint32_t format, const uint8_t kb[32], LiteMatch &g) {
    uint8_t payload[32];
    if (!payload_from_key(format, kb, payload)) return false;
    const std::string addr = address_from_payload(format, payload);
    if (addr.empty() || addr.size() >= sizeof g.address) return false;
    memcpy(g.key, kb, 32);
    memcpy(g.address, addr.c_str(), addr.size() + 1);
    return true;
}

// Threads for host-side work (candidate confirmation, rendering): the cores this process may really use — affinity mask,
// capped by a cgroup-v2 CPU quota —, not the machine's thread count (a 256-thread host with a 16-core quota ran 256 workers).
unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n ? n : 1024, (unsigned)CPU_COUNT(&set));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min<unsigned>(n, (unsigned)std::max(1L, atol(quota) / period));
        fclose(f);
    }
    return std::max(1u, std::min(n, 64u));
}

// Checkpoint of one scan (the reference has none): which batches of every shard are
// finished, and the matches found in them, so that an interrupted range / seeded scan resumes where it
// stopped instead of from its first key.  A batch is committed — its matches appended and its shard's
// counter advanced — under one lock, so every file written is a consistent prefix of the scan.  Shards
// process their batches in dispatch order, so "batches done" is a single number per shard.
//
// File (text, mode 0600 — it holds the private keys of the matches; rewritten atomically through <path>.tmp,
// fsync, rename):
//   vgen-hip checkpoint v1 / pattern_hex= / case_insensitive= / format= / batch_size= / n_shards= /
//   first_shard= / base= / end= / operations= / done=<per slot> / complete= / mode=range|random-seed24 / match=<key hex> ...
struct Checkpoint {
    std::string path;
    std::string pattern;
    int ci = 0;
    uint32_t format = 0, batch = 0, n_shards = 1, first_shard = 0;
    uint8_t base[32] = {0}, end[32] = {0};
    bool has_end = false;
    bool random = false;                   // a random-key scan: `base` holds its 24-byte seed (bytes 8..31), done[] counts batches of the streams
    double interval_s = 10.0;

    std::mutex mu;
    std::vector<uint64_t> done;            // per slot (slot = shard - first_shard)
    std::vector<LiteMatch> ledger;         // matches of committed batches, commit order
    uint64_t operations = 0;               // over all runs
    uint64_t resumed_operations = 0;       // as loaded
    bool complete = false;
    std::chrono::steady_clock::time_point last_write = std::chrono::steady_clock::now();
    std::string error;

    static std::string hex(const uint8_t *p, size_t n) { return hex_lower(p, n); }
    static bool unhex(const std::string &s, std::vector<uint8_t> &out) {
        if (s.size() % 2) return false;
        out.clear();
        for (size_t i = 0; i < s.size(); i += 2) {
            unsigned v;
            if (!isxdigit((unsigned char)s[i]) || !isxdigit((unsigned char)s[i + 1]) || sscanf(s.c_str() + i, "%2x", &v) != 1)
                return false;
            out.push_back((uint8_t)v);
        }
        return true;
    }

    // Loads `path` when it exists and checks that it describes this very scan.  `pin_base`: the caller
    // fixed the base key (config.start or a seed); otherwise the file's base key is adopted.
    // returns 1 = resumed, 0 = no file (fresh scan), -1 = error (see `error`).
    int load(bool pin_base) {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return 0;
        std::vector<std::pair<std::string, std::string>> kv;
        char line[4096];
        bool header = false;
        while (fgets(line, sizeof line, f)) {
            std::string s(line);
            while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
            if (!header) {
                if (s != "vgen-hip checkpoint v1") break;
                header = true;
                continue;
            }
            size_t eq = s.find('=');
            if (eq != std::string::npos) kv.emplace_back(s.substr(0, eq), s.substr(eq + 1));
        }
        fclose(f);
        if (!header) return bad("not a vgen-hip checkpoint file");
        auto get = [&](const char *k) -> const std::string * {
            for (auto &e : kv)
                if (e.first == k) return &e.second;
            return nullptr;
        };
        auto differs = [&](const char *k, const std::string &want) {
            const std::string *v = get(k);
            return !v || *v != want;
        };
        if (differs("pattern_hex", hex((const uint8_t *)pattern.data(), pattern.size()))) return bad("pattern");
        if (differs("case_insensitive", std::to_string(ci))) return bad("case_insensitive");
        if (differs("format", std::to_string(format))) return bad("format");
        if (differs("batch_size", std::to_string(batch))) return bad("batch_size");
        if (differs("n_shards", std::to_string(n_shards))) return bad("n_shards");
        if (differs("first_shard", std::to_string(first_shard))) return bad("first_shard");
        if (differs("end", has_end ? hex(end, 32) : "none")) return bad("end");
        {
            const std::string *m = get("mode");   // (files written before the field existed are key-range scans')
            // ("random" without the suffix: a file of round 3's 64-bit-seeded stream function, which no longer exists)
            if ((m ? *m : std::string("range")) != (random ? "random-seed24" : "range")) return bad("mode");
        }
        std::vector<uint8_t> b;
        const std::string *bs = get("base");
        if (!bs || !unhex(*bs, b) || b.size() != 32) return bad("base");
        if (pin_base && memcmp(b.data(), base, 32) != 0) return bad("base");
        memcpy(base, b.data(), 32);
        const std::string *d = get("done"), *o = get("operations"), *c = get("complete");
        if (!d || !o || !c) return bad("done/operations/complete");
        std::vector<uint64_t> dn;
        const char *p = d->c_str();
        while (*p) {
            char *e;
            dn.push_back(strtoull(p, &e, 10));
            if (e == p) return bad("done");
            p = e;
            while (*p == ' ') p++;
        }
        if (dn.size() != done.size()) return bad("done (slot count)");
        done = dn;
        operations = resumed_operations = strtoull(o->c_str(), nullptr, 10);
        complete = *c == "1";
        for (auto &e : kv) {
            if (e.first != "match") continue;
            LiteMatch g;
            if (!unhex(e.second, b) || b.size() != 32 || !lite_from_key(format, b.data(), g)) return bad("match");
            ledger.push_back(g);
        }
        return 1;
    }
    int bad(const char *field) {
        error = "checkpoint file '" + path + "' does not belong to this scan (" + field + ")";
        return -1;
    }

    // caller holds mu.  The file lists private keys (match=...): it is created 0600, never through a symlink,
    // and reaches the disk (fsync) before it replaces the previous checkpoint.
    bool write_locked() {
        const std::string tmp = path + ".tmp";
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0) return false;
        (void)fchmod(fd, 0600);   // an older .tmp may have been left with wider permissions
        FILE *f = fdopen(fd, "w");
        if (!f) {
            close(fd);
            return false;
        }
        fprintf(f, "vgen-hip checkpoint v1\npattern_hex=%s\ncase_insensitive=%d\nformat=%u\nbatch_size=%u\nn_shards=%u\n"
                   "first_shard=%u\nbase=%s\nend=%s\noperations=%llu\ndone=",
                hex((const uint8_t *)pattern.data(), pattern.size()).c_str(), ci, format, batch, n_shards, first_shard,
                hex(base, 32).c_str(), has_end ? hex(end, 32).c_str() : "none", (unsigned long long)operations);
        for (size_t i = 0; i < done.size(); i++) fprintf(f, "%s%llu", i ? " " : "", (unsigned long long)done[i]);
        fprintf(f, "\ncomplete=%d\nmode=%s\n", complete ? 1 : 0, random ? "random-seed24" : "range");
        for (auto &g : ledger) fprintf(f, "match=%s\n", hex(g.key, 32).c_str());
        bool ok = fflush(f) == 0 && fsync(fd) == 0;
        ok = (fclose(f) == 0) && ok;
        ok = ok && rename(tmp.c_str(), path.c_str()) == 0;
        last_write = std::chrono::steady_clock::now();
        return ok;
    }

    // One finished batch of `slot`: its matches and the shard's counter move together.
    void commit(uint32_t slot, const std::vector<LiteMatch> &batch_matches, uint64_t ops) {
        std::lock_guard<std::mutex> g(mu);
        ledger.insert(ledger.end(), batch_matches.begin(), batch_matches.end());
        done[slot]++;
        operations += ops;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - last_write).count() >= interval_s)
            (void)write_locked();
    }
};

// Host-side filtering of full dumps (the reference's rayon par_iter over every hash of a batch,
// src/gpu.rs:1030-1093): a pool of worker threads that lives as long as the scan, handed one index range per
// thread and batch.
class HostFilterPool {
public:
    explicit HostFilterPool(unsigned n) : n_(std::max(1u, n)) {
        for (unsigned t = 0; t < n_; t++) th_.emplace_back([this, t]() { loop(t); });
    }
    ~HostFilterPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    unsigned size() const { return n_; }
    // runs fn(t) on every worker t and returns when all are done
    void run(const std::function<void(unsigned)> &fn) {
        std::unique_lock<std::mutex> g(mu_);
        fn_ = &fn;
        pending_ = n_;
        gen_++;
        cv_.notify_all();
        done_.wait(g, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop(unsigned t) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *fn;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&]() { return gen_ != seen; });
                seen = gen_;
                if (quit_) return;
                fn = fn_;
            }
            (*fn)(t);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    unsigned n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned pending_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};

}  // namespace
}  // namespace vg

using namespace vg;

namespace {

// Where a shard (slot of the batch striping) stands: batches committed so far, in dispatch order.  A multi-device scan
// keeps one per slot so that a surviving context can take over the slot of a failed one exactly where it stopped
// (with a checkpoint the ledger's done[] is the same number and wins).
struct SlotProgress {
    std::atomic<uint64_t> done{0};
};

// One shard of a scan on one context.  `shared_found` (optional) is the match counter shared by the
// shards of a multi-device scan; without it the shard counts its own matches.  `ck` (optional): the
// scan's checkpoint; this shard is its slot `ck_slot`, skips the batches already recorded there and
// commits each batch it finishes.  `slot` (optional): the slot's progress — the first slot->done batches are skipped
// (another context committed them before it failed) and every batch committed here is counted in.
// *range_done: the shard stopped because its range ran out.
int scan_shard(vgen_ctx *ctx, const vgen_filter &flt, const vgen_scan_config *cfg, vgen_progress_cb cb, void *user,
               const volatile int32_t *stop, std::atomic<uint64_t> *shared_found, std::atomic<uint64_t> *shared_ops,
               MatchList &matches, uint64_t &total_ops, Checkpoint *ck = nullptr, uint32_t ck_slot = 0,
               bool *range_done = nullptr, SlotProgress *slot = nullptr, const RndSeed *scan_seed = nullptr) {
    if (cfg->format != ctx->format) return ctx->fail(VGEN_E_INVALID, "scan format differs from the context's format");
    const bool random_keys = (cfg->flags & VGEN_SCAN_RANDOM_KEYS) != 0;
    if (ctx->endo && !random_keys && (cfg->has_start || cfg->has_end || cfg->seed || cfg->n_shards > 1 || cfg->checkpoint_path))
        return ctx->fail(VGEN_E_INVALID, "a VGEN_FLAG_ENDO context tests six images of every point, not a contiguous key range: "
                                         "it serves unseeded random scans only (no start / end / seed / shards / checkpoint)");
    // (random keys on an endomorphism context: six keys per draw — the candidate and its lambda / negation images; seeds and
    //  shards keep their meaning there, they name streams of candidates, not ranges)
    if (random_keys && (cfg->has_start || cfg->has_end))
        return ctx->fail(VGEN_E_INVALID, "VGEN_SCAN_RANDOM_KEYS draws an independent key per candidate: no start / end");
    if (random_keys && ck && !scan_seed) return ctx->fail(VGEN_E_INVALID, "a checkpointed random-key scan needs its seed (open_checkpoint sets it)");

    const uint32_t N = ctx->batch;
    const size_t pbytes = (size_t)ctx->payload_words * 4;
    const uint32_t shards = cfg->n_shards > 1 ? cfg->n_shards : 1;
    const uint32_t shard = cfg->n_shards > 1 ? cfg->shard : 0;
    if (shard >= shards) return ctx->fail(VGEN_E_INVALID, "shard >= n_shards");

    // Device filter or host filtering of full dumps.  A prefilter whose expected candidates per batch do not fit the
    // match ring does not fall back to the reference's mode (every hash to the host): the ring GROWS to four times
    // the expectation, as long as that stays below half a record per key (beyond that nearly every key is a match
    // and the 20 B/key dump is the cheaper transfer).  A DEVF_DFA filter has no selectivity estimate: it starts on
    // the device with the ring it finds and adapts on overflow (below).
    auto next_pow2 = [](uint64_t v) {
        uint64_t p = 256;
        while (p < v) p <<= 1;
        return p;
    };
    bool host_all = flt.dev.kind == DEVF_HOST_ALL;
    const double keys_per_dispatch = (double)N * (ctx->endo ? 6 : 1);
    if (!host_all && flt.selectivity >= 0 && flt.selectivity * keys_per_dispatch * 4 > (double)ctx->match_cap) {
        const uint64_t want = next_pow2((uint64_t)(flt.selectivity * keys_per_dispatch * 4));
        if (want <= N / 2) {
            int rc = vgen_set_match_cap(ctx, (uint32_t)want);
            if (rc != VGEN_OK) return rc;
        } else {
            host_all = true;
        }
    }
    int rc = vgen_set_filter(ctx, host_all ? nullptr : &flt);
    if (rc != VGEN_OK) return rc;

    // Scans that multiply a scalar per key (random keys, taproot): how wide a generator table is this scan worth?  The default
    // 24-bit table (10 additions per multiplication) is there in 30 ms; the 27-bit signed one (9 additions, +5 %) in 60 ms and
    // 21.5 GB; the 29-bit signed one (8 additions, +12.5 %) takes 0.7 - 2.3 s and 138 of the device's 288 GB (profiles/r04_gtab_signed.txt):
    // from 3 s of expected scanning the first pays for itself several times over, from 30 s the second.
    // From the keys the scan can expect to test — the range, max_batches, or count / the filter's selectivity — at the path's rate.
    // This is a PREFERENCE: the caller bounds it (cfg->table_bits_max, vgen_params.table_bits / device_mem_budget_bytes), the runtime
    // checks it against the device's free memory, builds the wider table in the background while this loop dispatches on the one
    // it has, and takes it into use when it is complete (runtime.cpp) — the scan never waits for a table.
    if (random_keys || ctx->format == VGF_P2TR) {
        // (a count-limited scan whose pattern has no selectivity estimate — the whole DFA on the device — is taken for short)
        double keys = cfg->count == UINT64_MAX ? 1e30 : 0.0;
        if (!host_all && flt.selectivity > 0 && cfg->count != UINT64_MAX) keys = (double)cfg->count / flt.selectivity;
        if (host_all && cfg->count != UINT64_MAX) keys = (double)cfg->count * 64.0;   // (nearly every key matches)
        if (cfg->max_batches) keys = std::min(keys, (double)cfg->max_batches * keys_per_dispatch);
        if (cfg->has_end && cfg->has_start) {
            Scalar a, b;
            scalar_from_be(a, cfg->start);
            scalar_from_be(b, cfg->end);
            bool small = true;     // the range fits 64 bits of distance?
            for (int i = 2; i < 8; i++) small = small && a.w[i] == b.w[i];
            if (small) {
                const uint64_t lo = (uint64_t)a.w[1] << 32 | a.w[0], hi = (uint64_t)b.w[1] << 32 | b.w[0];
                if (hi >= lo) keys = std::min(keys, (double)(hi - lo) + 1.0);
            }
        }
        const double seconds = keys / (random_keys && ctx->endo ? 5.5e9 : 1.35e9) / (double)shards;
        rt_prefer_table_bits(ctx, seconds >= 30.0 ? 29u : seconds >= 3.0 ? 27u : 0u, cfg->table_bits_max);
    }

    // independent random keys: candidate index = batch number x N within stream `shard` of the seed
    // (the seed of the whole scan when the caller resolved one — all shards of a multi-device scan and a resumed checkpoint
    //  share it —, else this call's own)
    const RndSeed rnd_seed = !random_keys ? RndSeed{} : scan_seed ? *scan_seed : scan_rnd_seed(cfg->seed);
    Scalar current;
    if (random_keys) {
        memset(&current, 0, sizeof current);
        current.w[0] = 1;   // (unused: keeps the range bookkeeping below on valid ground)
    } else if (cfg->has_start) {
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
    if (shard && !random_keys) exhausted = scalar_add_u64(current, current, (uint64_t)shard * N) || !scalar_is_valid(current);
    const uint64_t stride = (uint64_t)shards * N;
    // resume / take-over: this shard's first `skipped` batches are already in the checkpoint, or were committed by the
    // context that owned the slot before it failed
    const uint64_t skipped = ck ? ck->done[ck_slot] : slot ? slot->done.load() : 0;
    const uint64_t taken_over = slot && !ck ? skipped : 0;   // counts against max_batches: the slot's budget, not the context's
    if (skipped && !exhausted && !random_keys) {
        uint64_t skip = skipped;
        while (skip && !exhausted) {
            const uint64_t step = std::min<uint64_t>(skip, UINT64_MAX / stride);
            exhausted = scalar_add_u64(current, current, step * stride) || !scalar_is_valid(current);
            skip -= step;
        }
    }

    uint64_t dispatched = 0;
    total_ops = 0;
    const uint64_t count = cfg->count;
    auto found = [&]() -> uint64_t { return shared_found ? shared_found->load(std::memory_order_relaxed) : matches.size(); };
    std::vector<LiteMatch> batch_matches;        // matches of the batch being processed (checkpoint commit unit)
    // A confirmed match: into the result while `count` is not reached; with a checkpoint ALWAYS into the batch's
    // ledger entry, so that a committed batch is recorded with all of its matches and a later run with a larger
    // count loses none.  Returns false when the match was dropped (no checkpoint, count reached).
    uint64_t taken_uncommitted = 0;   // matches taken from the batch in hand, not yet committed (rolled back if the scan fails first)
    auto push = [&](const LiteMatch &g) -> bool {
        const bool take = found() < count;
        if (take) {
            matches.push_back(g);
            taken_uncommitted++;
            if (shared_found) shared_found->fetch_add(1, std::memory_order_relaxed);
        }
        if (ck) batch_matches.push_back(g);
        return take || ck;
    };
    // The workers' confirmed matches of one batch (index order: part 0, part 1, ...), handed over in bulk: as many as `count`
    // still has room for, all of them into the checkpoint's batch record.  Returns true when some were dropped.
    auto push_parts = [&](std::vector<std::vector<LiteMatch>> &part) -> bool {
        size_t total = 0;
        for (auto &p : part) total += p.size();
        const uint64_t have = found();
        const uint64_t room = have < count ? count - have : 0;
        size_t take = (size_t)std::min<uint64_t>(total, room);
        const size_t taken = take;
        for (auto &p : part) {
            if (ck) batch_matches.insert(batch_matches.end(), p.begin(), p.end());
            const size_t k = std::min(take, p.size());
            matches.take(std::move(p), k);     // the worker's vector itself becomes a block of the result: nothing is copied
            take -= k;
        }
        taken_uncommitted += taken;
        if (shared_found && taken) shared_found->fetch_add(taken, std::memory_order_relaxed);
        return taken < total && !ck;
    };
    // frames this scan drives: all of them, or — filtering full dumps on the host — those that have a dump buffer
    // (runtime.cpp: ensure_dump_slab bounds the pinned memory; the host filter is the bottleneck there anyway)
    uint32_t nf = ctx->frames;
    if (host_all && ctx->dump_frames) nf = std::min(nf, ctx->dump_frames);
    std::vector<BatchKeys> pend(ctx->frames);
    uint32_t in_flight = 0;
    int status = VGEN_OK;

    // (the host's stop flag is written by another thread: an atomic read, src/gpu.rs:980-984 reads its AtomicBool Relaxed)
    auto stopped = [&]() { return stop && __atomic_load_n(const_cast<const int32_t *>(stop), __ATOMIC_RELAXED) != 0; };
    auto in_range = [&]() { return !exhausted && (!end || scalar_cmp(current, *end) <= 0); };
    auto can_dispatch = [&]() { return in_range() && (!cfg->max_batches || dispatched + taken_over < cfg->max_batches); };
    auto dispatch = [&](uint32_t frame) -> int {
        if (random_keys) {
            // every shard owns a stream of its own and walks its candidates in order (the oracle's worker thread, oracle/vo_scan.c)
            const uint64_t batch_no = skipped + dispatched;
            const uint64_t first = batch_no * (uint64_t)N;
            if (first / N != batch_no || first > UINT64_MAX - (N - 1)) {
                exhausted = true;
                return ctx->fail(VGEN_E_RANGE, "random-key stream exhausted (2^64 candidates)");
            }
            uint8_t sb[24];
            rnd_seed_to_bytes(rnd_seed, sb);
            int r = vgen_dispatch_random_seed(ctx, frame, sb, shard, first);
            if (r != VGEN_OK) return r;
            pend[frame] = BatchKeys{};
            pend[frame].random = true;
            pend[frame].seed = rnd_seed;
            pend[frame].stream = shard;
            pend[frame].first_index = first;
            dispatched++;
            return VGEN_OK;
        }
        uint8_t kb[32];
        scalar_to_be(current, kb);
        int r = vgen_dispatch(ctx, frame, kb);
        if (r != VGEN_OK) return r;
        pend[frame] = BatchKeys{};
        pend[frame].start = current;
        dispatched++;
        Scalar nx;
        if (scalar_add_u64(nx, current, stride) || !scalar_is_valid(nx)) exhausted = true;   // key space exhausted
        else current = nx;
        return VGEN_OK;
    };

    // Frames in flight, in dispatch order (results are consumed in that order, so matches stay in ascending batch
    // order).  The scan starts with two frames and activates one more per completed batch: with all frames primed
    // at once the first result would wait for its share of a device busy with sixteen dispatches (2.3 ms to the
    // first match instead of ~0.3 ms); a long scan reaches the full depth after `frames` batches.
    std::deque<uint32_t> order;
    uint32_t active = 0;   // frames 0 .. active-1 have been put to use
    auto launch = [&](uint32_t f) -> bool {
        if ((status = dispatch(f)) != VGEN_OK) return false;
        order.push_back(f);
        in_flight++;
        return true;
    };
    // A scan of the scalar-multiplication paths that turns out LONG although nothing said so up front (no selectivity estimate, a
    // count that keeps not being reached): after 5 s it asks for the 29-bit signed table (+12.5 % once it is there) and simply goes
    // on — the runtime builds it behind the dispatches (round 4 drained the frames and paused 0.7 - 2.3 s here).
    const bool table_path = random_keys || ctx->format == VGF_P2TR;
    const auto scan_t0 = std::chrono::steady_clock::now();
    bool upgrade_asked = false;
    auto may_launch = [&]() { return can_dispatch() && !stopped() && found() < count; };
    // (a frame's stream — a hardware queue of its own — is created at its first dispatch and takes ~8 ms: a fresh context
    // starts on frame 0 alone, so that an easy pattern's first match does not wait for a second queue it never needs)
    auto prime = [&]() {
        while (active < std::min<uint32_t>(nf, 2) && may_launch()) {
            if (active >= 1 && !rt_frame_ready(ctx, active)) break;
            if (!launch(active++)) break;
        }
    };
    prime();

    std::vector<vgen_match> recs(ctx->match_cap);   // (after the ring has its size for this scan)
    std::unique_ptr<HostFilterPool> pool;   // created with the first dumped batch / the first batch with thousands of candidates
    bool cut_any = false;      // (no checkpoint) some batch had matches beyond `count` dropped: the range was not covered

#ifdef VGEN_SCAN_PROFILE
    double prof_wait = 0, prof_pool = 0, prof_merge = 0, prof_dispatch = 0;
    uint64_t prof_cand = 0;
#endif
    while (status == VGEN_OK && !order.empty()) {
        const uint32_t frame = order.front();
        order.pop_front();
        uint32_t n_found = 0;
        uint64_t tested = 0;
        PROF(prof_wait, status = vgen_wait(ctx, frame, recs.data(), (uint32_t)recs.size(), &n_found, &tested));
        if (status != VGEN_OK) break;
        in_flight--;
        const BatchKeys batch_start = pend[frame];
        const bool dumped = ctx->fr[frame].dumped;
        const uint32_t images = tested > N ? (uint32_t)(tested / N) : 1;   // 6 for an endomorphism dispatch
        bool cut = false;      // this batch: a confirmed (or unexamined) match was dropped because `count` was reached

        if (dumped) {
            // Host filtering of the whole batch (the reference's only mode, gpu.rs:1030-1093) straight from the
            // pinned buffer the dispatch copied itself into — hence BEFORE the frame is dispatched again —, in
            // parallel index ranges, results kept in ascending index order.  The device is not the bottleneck
            // here (encoding and matching 2^20 addresses takes the host tens of milliseconds).
            const uint8_t *dump = nullptr;
            if ((status = vgen_dump_view(ctx, frame, &dump, nullptr)) != VGEN_OK) break;
            // A scan that wants few matches of a pattern nearly every key satisfies (the reference's default `range
            // --puzzle N`: pattern ".", count 1, src/lib.rs:519) must not encode a million addresses to return the first:
            // the dump is examined in index order in growing pieces — a first short one on this thread — until `count` is
            // reached; what is left unexamined counts as cut.  With a checkpoint the whole batch is examined (a committed
            // batch is recorded with every match it holds).
            const uint32_t total = (uint32_t)tested;   // N, or 6 N for an endomorphism dispatch
            uint32_t pos = 0;
            for (unsigned round = 0; pos < total && (ck || found() < count); round++) {
                const uint64_t have = found();
                const uint64_t need = ck ? total : have < count ? count - have : 0;
                uint64_t len = total - pos;
                if (!ck && need < total / 8) len = std::min<uint64_t>(len, std::max<uint64_t>(need * 4, 64) << std::min(2 * round, 24u));
                if (len < 1024) {
                    LiteMatch g;
                    const uint32_t stop_at = pos + (uint32_t)len;
                    for (; pos < stop_at && (ck || found() < count); pos++)
                        if (make_match(flt, cfg->format, batch_start, pos, dump + (size_t)pos * pbytes, end, g, N, images)) (void)push(g);
                    continue;
                }
                if (!pool) pool.reset(new HostFilterPool(host_threads()));
                const unsigned nt = pool->size();
                std::vector<std::vector<LiteMatch>> part(nt);
                const uint32_t base = pos, span = (uint32_t)len;
                pool->run([&](unsigned t) {
                    const uint32_t lo = base + (uint32_t)((uint64_t)span * t / nt), hi = base + (uint32_t)((uint64_t)span * (t + 1) / nt);
                    LiteMatch g;
                    for (uint32_t i = lo; i < hi; i++)
                        if (make_match(flt, cfg->format, batch_start, i, dump + (size_t)i * pbytes, end, g, N, images)) part[t].push_back(g);
                });
                if (push_parts(part)) cut = true;
                pos += span;
            }
            if (pos < total) cut = true;   // keys left unexamined (conservative: they may not all be matches)
        }

        bool dispatched_next = false;
        bool ramp_later = false;
        if (may_launch()) {
            bool ok_launch = true;
            PROF(prof_dispatch, ok_launch = launch(frame));
            if (!ok_launch) break;
            dispatched_next = true;
            // ramp up: one more frame per batch — at once when its stream exists; frames that still need their stream
            // (a hardware queue, ~8 ms to create) join as a helper thread gets the streams made, which is asked for
            // after this batch's candidates have been examined and only if the scan goes on
            if (active < nf && may_launch()) {
                if (!rt_frame_ready(ctx, active)) ramp_later = true;
                else if (!launch(active++)) break;
            }
        }

        if (!dumped) {
            if (n_found > recs.size()) {
                // More candidates than the ring holds (a permissive pattern): nothing may be dropped, so drain
                // what is in flight, grow the ring (x4, at least twice what this batch produced) — or, once a ring
                // would need more than half a record per key, switch to host filtering of full dumps, the
                // reference's mode — and redo from this batch.
                for (uint32_t f = 0; f < ctx->frames; f++)
                    if (ctx->fr[f].in_flight) (void)vgen_wait(ctx, f, nullptr, 0, nullptr, nullptr);
                const uint64_t want = next_pow2(std::max<uint64_t>((uint64_t)recs.size() * 4, (uint64_t)n_found * 2));
                if (want <= N / 2) {
                    if ((status = vgen_set_match_cap(ctx, (uint32_t)want)) != VGEN_OK) break;
                    recs.resize(ctx->match_cap);
                } else if ((status = vgen_set_filter(ctx, nullptr)) != VGEN_OK) {
                    break;
                } else if (ctx->dump_frames) {
                    nf = std::min(nf, ctx->dump_frames);
                }
                dispatched -= 1 + in_flight;
                in_flight = 0;
                order.clear();
                active = 0;
                if (!random_keys) current = batch_start.start;
                exhausted = false;
                prime();
                continue;
            }
            LiteMatch g;
            uint32_t i = 0;
            if (n_found >= 2048) {
                // many candidates (a permissive pattern on a grown ring): confirm them on the worker pool, in index order
                if (!pool) pool.reset(new HostFilterPool(host_threads()));
                const unsigned nt = pool->size();
                std::vector<std::vector<LiteMatch>> part(nt);
                PROF(prof_pool, pool->run([&](unsigned t) {
                    const uint32_t lo = (uint32_t)((uint64_t)n_found * t / nt), hi = (uint32_t)((uint64_t)n_found * (t + 1) / nt);
                    LiteMatch gg;
                    for (uint32_t k = lo; k < hi; k++)
                        if (make_match(flt, cfg->format, batch_start, recs[k].index, recs[k].payload, end, gg, N, images)) part[t].push_back(gg);
                }));
                PROF(prof_merge, if (push_parts(part)) cut = true;);
#ifdef VGEN_SCAN_PROFILE
                prof_cand += n_found;
#endif
            } else {
                for (; i < n_found && (ck || found() < count); i++)
                    if (make_match(flt, cfg->format, batch_start, recs[i].index, recs[i].payload, end, g, N, images)) (void)push(g);
                cut = i < n_found;   // candidates left unexamined (conservative: they may not all be matches)
            }
        }

        total_ops += tested;                         // gpu.rs:1106 (batch_size; six times that for an endomorphism dispatch)
        // With a checkpoint a batch is committed with every match it holds (push above), also those beyond
        // `count`: the file stays a consistent prefix of the scan whatever count a later run asks for.
        cut_any = cut_any || cut;
        if (ck) ck->commit(ck_slot, batch_matches, tested);
        if (slot) slot->done.fetch_add(1);
        taken_uncommitted = 0;
        batch_matches.clear();
        if (cb) cb(shared_ops ? tested : total_ops, user);   // multi-device: the wrapper adds N to the shared count under its lock
        if (found() >= count && !dispatched_next) break;   // gpu.rs:1111
        if (table_path && !upgrade_asked && std::chrono::duration<double>(std::chrono::steady_clock::now() - scan_t0).count() >= 5.0) {
            rt_prefer_table_bits(ctx, 29, cfg->table_bits_max);   // taken into use when it is complete; nothing waits
            upgrade_asked = true;
        }
        if (ramp_later && active < nf && may_launch()) {
            // not ready and no helper for this stream kind: create it here and now, as before
            if (!rt_frame_ready(ctx, active) && rt_prepare_streams(ctx)) continue;
            if (!launch(active++)) break;
        }
    }
#ifdef VGEN_SCAN_PROFILE
    fprintf(stderr, "[scan profile] wait %.3f s  pool %.3f s  merge %.3f s  dispatch %.3f s  candidates %llu  matches %zu  ops %llu\n", prof_wait, prof_pool,
            prof_merge, prof_dispatch, (unsigned long long)prof_cand, matches.size(), (unsigned long long)total_ops);
#endif
    // A scan that fails between examining a batch and committing it (in dump mode the host filter runs BEFORE the frame is
    // dispatched again, and that dispatch is where a dead device shows) must not keep that batch's matches: the batch is not
    // counted as done, so whoever resumes or takes over the slot will produce them again.
    if (status != VGEN_OK && taken_uncommitted) {
        matches.truncate(matches.size() - (size_t)taken_uncommitted);
        if (shared_found) shared_found->fetch_sub(taken_uncommitted, std::memory_order_relaxed);
    }
    // drain anything still in flight (the reference drops its runner; we must not leave frames busy)
    const bool all_processed = order.empty() && !cut_any;   // no dispatched batch was left unread or cut short
    for (uint32_t f = 0; f < ctx->frames; f++)
        if (ctx->fr[f].in_flight) (void)vgen_wait(ctx, f, nullptr, 0, nullptr, nullptr);
    if (range_done) *range_done = status == VGEN_OK && !in_range() && all_processed;
    return status;
}

// The base key of a scan without config.start: seeded or drawn from OS entropy.
void resolve_base(vgen_scan_config &c) {
    if (c.has_start) return;
    Scalar k;
    if (c.seed) seed_key(c.seed, 0, k);
    else random_valid_key(k);
    scalar_to_be(k, c.start);
    c.has_start = 1;
}

// Sets up the checkpoint of a scan over `slots` shards starting at shard `first_shard`; resumes from the
// file when there is one (adopting its base key for an unseeded random scan).  VGEN_OK / error.
int open_checkpoint(vgen_ctx *ctx, Checkpoint &ck, const char *pattern, vgen_scan_config &c, uint32_t batch, uint32_t n_shards,
                    uint32_t first_shard, uint32_t slots, RndSeed &rnd_seed) {
    const bool random_keys = (c.flags & VGEN_SCAN_RANDOM_KEYS) != 0;
    const bool pin_base = c.has_start || c.seed;
    if (random_keys) {
        // the scan is named by its seed (OS entropy when the caller gave none: the file then carries it to the next run);
        // `done` counts the batches of each stream
        rnd_seed = scan_rnd_seed(c.seed);
        memset(c.start, 0, 32);
        rnd_seed_to_bytes(rnd_seed, c.start + 8);
    } else {
        resolve_base(c);
    }
    ck.random = random_keys;
    ck.path = c.checkpoint_path;
    ck.pattern = pattern;
    ck.ci = c.case_insensitive != 0;
    ck.format = c.format;
    ck.batch = batch;
    ck.n_shards = n_shards;
    ck.first_shard = first_shard;
    memcpy(ck.base, c.start, 32);
    ck.has_end = c.has_end != 0;
    if (ck.has_end) memcpy(ck.end, c.end, 32);
    if (c.checkpoint_interval_ms) ck.interval_s = c.checkpoint_interval_ms / 1000.0;
    ck.done.assign(slots, 0);
    const int r = ck.load(pin_base);
    if (r < 0) return ctx->fail(VGEN_E_INVALID, ck.error);
    memcpy(c.start, ck.base, 32);
    if (random_keys) {   // (the file's seed when the caller gave none)
        rnd_seed = rnd_seed_from_bytes(ck.base + 8);
        memset(c.start, 0, 32);
    }
    return VGEN_OK;
}

// Hands the matches to the caller as GeneratedAddress records: this is where WIF and hex are rendered — for the matches that
// survived `count`, straight into the result array, on several threads when there are thousands.
int finish_result(vgen_ctx *ctx, uint32_t format, const MatchList &matches, uint64_t ops, double secs, vgen_scan_result *out) {
    out->n_matches = matches.size();
    out->operations = ops;
    if (!matches.empty()) {
        const auto t0 = std::chrono::steady_clock::now();
        out->matches = (vgen_generated *)malloc(matches.size() * sizeof(vgen_generated));
        if (!out->matches) return ctx->fail(VGEN_E_NOMEM, "out of memory");
        const size_t n = matches.size();
        const unsigned nt = n >= 4096 ? host_threads() : 1;
        const auto &blocks = matches.blocks();
        auto work = [&](unsigned t) {
            // entries [lo, hi) of the list, walked block by block
            const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
            size_t base = 0;
            for (auto &b : blocks) {
                const size_t from = std::max(lo, base), to = std::min(hi, base + b.size());
                for (size_t i = from; i < to; i++) render_match(format, b[i - base], out->matches[i]);
                base += b.size();
                if (base >= hi) break;
            }
        };
        if (nt == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();   // rendering is part of the scan's time
    }
    out->elapsed_secs = secs;
    return VGEN_OK;
}

int finish_result(vgen_ctx *ctx, uint32_t format, std::vector<LiteMatch> &matches, uint64_t ops, double secs, vgen_scan_result *out) {
    MatchList l;
    l.take(std::move(matches), matches.size());
    return finish_result(ctx, format, l, ops, secs, out);
}

}  // namespace

// ABI 4's vgen_scan_config, or ABI 3's 136 bytes (no table_bits_max: no cap): -> the full structure, fields beyond the caller's zero.
static bool normalise_scan_config(const vgen_scan_config *in, vgen_scan_config &full) {
    constexpr uint32_t CONFIG_ABI3 = 136;
    if (!in || (in->struct_size != sizeof(vgen_scan_config) && in->struct_size != CONFIG_ABI3)) return false;
    memset(&full, 0, sizeof full);
    memcpy(&full, in, in->struct_size);
    full.struct_size = sizeof full;
    return true;
}

extern "C" int vgen_scan(vgen_ctx *ctx, const char *pattern, const vgen_scan_config *cfg_in, vgen_progress_cb cb,
                         void *user, const volatile int32_t *stop, vgen_scan_result *out) {
    vgen_scan_config cfg_full;
    if (!ctx || !pattern || !out || !normalise_scan_config(cfg_in, cfg_full)) return VGEN_E_INVALID;
    const vgen_scan_config *cfg = &cfg_full;
    memset(out, 0, sizeof *out);
    const auto t0 = std::chrono::steady_clock::now();
    vgen_filter flt;
    std::string err;
    if (!filter_compile(pattern, cfg->case_insensitive != 0, cfg->format, flt, err))
        return ctx->fail(VGEN_E_PATTERN, err);
    MatchList matches;
    uint64_t ops = 0;
    bool range_done = false;
    if (!cfg->checkpoint_path) {
        int rc = scan_shard(ctx, flt, cfg, cb, user, stop, nullptr, nullptr, matches, ops, nullptr, 0, &range_done);
        if (rc != VGEN_OK) {
            // the error, AND what the batches finished before it had found (complete = 0): a host that falls back to
            // another backend (the reference's run_search does, src/lib.rs:727-746,1185-1198) keeps those matches
            const std::string why = ctx->err;
            (void)finish_result(ctx, cfg->format, matches, ops, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
            out->failed_shards = 1;
            ctx->err = why;
            return rc;
        }
        out->complete = range_done;
        return finish_result(ctx, cfg->format, matches, ops, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
    }
    vgen_scan_config c = *cfg;
    Checkpoint ck;
    const uint32_t shards = c.n_shards > 1 ? c.n_shards : 1;
    RndSeed rnd_seed{};
    int rc = open_checkpoint(ctx, ck, pattern, c, ctx->batch, shards, c.n_shards > 1 ? c.shard : 0, 1, rnd_seed);
    if (rc != VGEN_OK) return rc;
    matches.append_copy(ck.ledger);   // what earlier runs found counts towards `count` (committed batches keep all their matches)
    if (matches.size() > c.count) matches.truncate((size_t)c.count);
    if (!ck.complete && matches.size() < c.count)
        rc = scan_shard(ctx, flt, &c, cb, user, stop, nullptr, nullptr, matches, ops, &ck, 0, &range_done, nullptr, &rnd_seed);
    {
        std::lock_guard<std::mutex> g(ck.mu);
        ck.complete = ck.complete || (rc == VGEN_OK && range_done);
        if (!ck.write_locked() && rc == VGEN_OK) rc = ctx->fail(VGEN_E_INVALID, "cannot write checkpoint file '" + ck.path + "'");
    }
    if (rc != VGEN_OK) {
        const std::string why = ctx->err;
        std::vector<LiteMatch> all;
        {
            std::lock_guard<std::mutex> g(ck.mu);
            all = ck.ledger;   // every batch committed before the failure, earlier runs included (the file holds the same)
        }
        if (all.size() > c.count) all.resize((size_t)c.count);
        (void)finish_result(ctx, cfg->format, all, ops, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
        out->resumed_operations = ck.resumed_operations;
        out->failed_shards = 1;
        ctx->err = why;
        return rc;
    }
    out->complete = ck.complete;
    out->resumed_operations = ck.resumed_operations;
    return finish_result(ctx, cfg->format, matches, ops, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
}

// Multi-device scan: one host thread per context, batches striped over the contexts (context i takes
// global batches b = i mod n), a shared match counter and stop flag, results merged in ascending key
// order and truncated to `count` (no collective, host-side aggregation only).
extern "C" int vgen_scan_multi(vgen_ctx **ctxs, uint32_t n_ctx, const char *pattern, const vgen_scan_config *cfg_in,
                               vgen_progress_cb cb, void *user, const volatile int32_t *stop, vgen_scan_result *out) {
    vgen_scan_config cfg_full;
    if (!ctxs || n_ctx == 0 || !pattern || !out || !normalise_scan_config(cfg_in, cfg_full)) return VGEN_E_INVALID;
    const vgen_scan_config *cfg = &cfg_full;
    for (uint32_t i = 0; i < n_ctx; i++)
        if (!ctxs[i] || ctxs[i]->batch != ctxs[0]->batch) return VGEN_E_INVALID;
    memset(out, 0, sizeof *out);
    const auto t0 = std::chrono::steady_clock::now();
    vgen_filter flt;
    std::string err;
    if (!filter_compile(pattern, cfg->case_insensitive != 0, cfg->format, flt, err))
        return ctxs[0]->fail(VGEN_E_PATTERN, err);
    vgen_scan_config base = *cfg;
    Checkpoint ck;
    Checkpoint *ckp = nullptr;
    // Endomorphism contexts test images of points, not a range to stripe: every device walks from a random base of its
    // own (disjoint with overwhelming probability, SURVEY.md 8(e) "random mode"), sharing the match counter.
    bool endo = false;
    for (uint32_t i = 0; i < n_ctx; i++) endo = endo || ctxs[i]->endo;
    const bool random_keys = (cfg->flags & VGEN_SCAN_RANDOM_KEYS) != 0;
    if (endo) {
        for (uint32_t i = 0; i < n_ctx; i++)
            if (!ctxs[i]->endo) return ctxs[0]->fail(VGEN_E_INVALID, "vgen_scan_multi: VGEN_FLAG_ENDO must be set on all contexts or on none");
        if (!random_keys && (cfg->has_start || cfg->has_end || cfg->seed || cfg->checkpoint_path))
            return ctxs[0]->fail(VGEN_E_INVALID, "VGEN_FLAG_ENDO contexts serve unseeded random scans only (no start / end / seed / checkpoint)");
    }
    // the walk of an endomorphism context starts from a random base of its own per device; random-key scans stripe by
    // stream (shard i walks stream i) whatever the context
    const bool own_bases = endo && !random_keys;   // no slots to stripe or adopt: every context walks from its own random base
    RndSeed rnd_seed{};   // random-key scans: one seed for all streams (shard i walks stream i of it)
    if (cfg->checkpoint_path) {
        int rc = open_checkpoint(ctxs[0], ck, pattern, base, ctxs[0]->batch, n_ctx, 0, n_ctx, rnd_seed);
        if (rc != VGEN_OK) return rc;
        ckp = &ck;
    } else if (random_keys) {
        rnd_seed = scan_rnd_seed(cfg->seed);
    } else if (!own_bases) {
        resolve_base(base);   // all shards must walk the same base key
    }
    std::atomic<uint64_t> found{0}, ops_shared{0};
    if (ckp) found = ck.ledger.size();
    const bool skip_all = ckp && (ck.complete || ck.ledger.size() >= cfg->count);
    // Per SLOT of the striping (slot i starts on context i): matches, operations, progress, whether its range ran out.
    // Per CONTEXT: the status of its last scan_shard.  A context that fails retires; its slot is left where its last
    // committed batch put it, and a context that has finished its own slot takes it over from there (the reference has one
    // adapter and falls back to its CPU path instead, src/lib.rs:727-746,1185-1198; SURVEY.md 5: "per-GPU worker failure =>
    // re-queue its range on surviving GPUs").  Batches in flight on the failed context were never committed: the adopter
    // redoes them.  Contexts that finish while others are still running wait for a possible orphan instead of exiting.
    std::vector<MatchList> part(n_ctx);
    std::vector<uint64_t> ops(n_ctx, 0);
    std::vector<int> rcs(n_ctx, VGEN_OK);
    std::vector<char> range_done(n_ctx, 0), slot_finished(n_ctx, 0);
    std::vector<SlotProgress> progress(n_ctx);
    std::mutex cb_mu, q_mu;
    std::condition_variable q_cv;
    std::deque<uint32_t> orphans;
    uint32_t running = skip_all ? 0 : n_ctx;   // threads still inside a scan_shard call
    uint32_t failed_ctx = 0;
    // one callback per finished batch of any shard, with the cumulative count of all shards: the increment and
    // the call share a lock, so the host sees strictly increasing multiples of the batch size (gpu.rs:1106-1109)
    struct CbCtx { vgen_progress_cb cb; void *user; std::mutex *mu; std::atomic<uint64_t> *ops; } cbc{cb, user, &cb_mu, &ops_shared};
    auto locked_cb = [](uint64_t delta, void *u) {
        CbCtx *c = (CbCtx *)u;
        std::lock_guard<std::mutex> g(*c->mu);
        c->cb(c->ops->fetch_add(delta) + delta, c->user);
    };
    auto scan_over = [&]() {
        return (stop && __atomic_load_n(const_cast<const int32_t *>(stop), __ATOMIC_RELAXED) != 0) || found.load() >= cfg->count;
    };
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n_ctx && !skip_all; i++)
        th.emplace_back([&, i]() {
            uint32_t slot = i;
            for (;;) {
                vgen_scan_config c = base;
                c.shard = own_bases ? 0 : slot;
                c.n_shards = own_bases ? 0 : n_ctx;
                bool rd = false;
                MatchList got;
                uint64_t o = 0;
                const int rc = scan_shard(ctxs[i], flt, &c, cb ? (vgen_progress_cb)locked_cb : nullptr, &cbc, stop, &found, &ops_shared,
                                          got, o, ckp, slot, &rd, own_bases ? nullptr : &progress[slot], random_keys ? &rnd_seed : nullptr);
                std::unique_lock<std::mutex> lk(q_mu);
                part[slot].append(std::move(got));
                ops[slot] += o;
                if (rc != VGEN_OK) {
                    // this context retires; the slot it was working on is up for adoption (never for endomorphism contexts:
                    // they walk from random bases of their own, there is no range to complete)
                    rcs[i] = rc;
                    failed_ctx++;
                    running--;
                    if (!own_bases) orphans.push_back(slot);
                    q_cv.notify_all();
                    return;
                }
                range_done[slot] = rd;
                slot_finished[slot] = 1;
                // finished a slot: adopt an orphan if the scan still wants keys, else wait while anybody may still fail
                running--;
                q_cv.notify_all();
                q_cv.wait(lk, [&]() { return !orphans.empty() || running == 0 || scan_over(); });
                if (orphans.empty() || scan_over()) return;
                slot = orphans.front();
                orphans.pop_front();
                running++;
            }
        });
    for (auto &x : th) x.join();
    bool all_done = !skip_all;
    for (uint32_t i = 0; i < n_ctx; i++) all_done = all_done && slot_finished[i] && range_done[i];
    int first_err = VGEN_OK;
    uint32_t first_err_ctx = 0;
    for (uint32_t i = 0; i < n_ctx; i++)
        if (rcs[i] != VGEN_OK && first_err == VGEN_OK) {
            first_err = rcs[i];
            first_err_ctx = i;
        }
    // Every slot covered (by its own context or an adopter), or the scan ended because `count` / the stop flag said so:
    // the failures were absorbed.  Otherwise (no context left to adopt a slot) the scan is incomplete and the call fails —
    // still handing over everything the committed batches found.
    bool uncovered = false;
    for (uint32_t i = 0; i < n_ctx && !skip_all; i++) uncovered = uncovered || !slot_finished[i];
    const bool absorbed = first_err != VGEN_OK && (!uncovered || scan_over() || own_bases) && failed_ctx < n_ctx;
    if (ckp) {
        std::lock_guard<std::mutex> g(ck.mu);
        ck.complete = ck.complete || all_done;
        if (!ck.write_locked() && first_err == VGEN_OK) {
            first_err = ctxs[0]->fail(VGEN_E_INVALID, "cannot write checkpoint file '" + ck.path + "'");
            first_err_ctx = 0;
        }
        out->resumed_operations = ck.resumed_operations;
    }
    out->complete = ckp ? ck.complete : all_done;
    out->failed_shards = (int32_t)failed_ctx;
    std::vector<LiteMatch> all;
    uint64_t total = 0;
    if (ckp) all = ck.ledger;   // earlier runs' matches + every batch committed by this one
    for (uint32_t i = 0; i < n_ctx; i++) {
        if (!ckp) {
            const std::vector<LiteMatch> flat = part[i].flatten();
            all.insert(all.end(), flat.begin(), flat.end());
        }
        total += ops[i];
    }
    std::sort(all.begin(), all.end(), [](const LiteMatch &a, const LiteMatch &b) { return memcmp(a.key, b.key, 32) < 0; });
    if (all.size() > cfg->count) all.resize((size_t)cfg->count);
    const std::string why = first_err != VGEN_OK ? ctxs[first_err_ctx]->err : std::string();
    const int frc = finish_result(ctxs[0], cfg->format, all, total, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
    if (first_err != VGEN_OK && !absorbed) {
        out->complete = 0;
        ctxs[first_err_ctx]->err = why;
        if (first_err_ctx != 0) ctxs[0]->err = "context " + std::to_string(first_err_ctx) + ": " + why;
        return first_err;
    }
    if (first_err != VGEN_OK) ctxs[first_err_ctx]->err = why;   // absorbed: still readable through vgen_last_error(ctxs[i])
    return frc;
}

extern "C" void vgen_scan_result_free(vgen_scan_result *r) {
    if (!r) return;
    free(r->matches);
    memset(r, 0, sizeof *r);
}
