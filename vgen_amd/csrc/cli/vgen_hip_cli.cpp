// vgen_hip_cli.cpp — `vgen-hip`: a thin command line over the C ABI of libvgen_hip.so, reproducing the
// flag set, defaults and result writers of the reference's `generate` / `range` / `list-gpus` /
// `verify` commands (src/lib.rs:44-211 flags, :642-663 range parsing, :524 count 0 = unbounded,
// :825-865 --repeat, :879-974 writers, :1038-1086 formatting helpers).  SURVEY.md §8(f) item 1.
//
// `estimate` (lib.rs:345-375) reports the device rate.  Provider patterns resolve against a static table (host/provider.cpp), not the boha crate.  Deliberately absent: the TUI and any CPU scan path (`--no-gpu` is an
// error here: this build has no CPU backend).  Added: --seed (the reference seeds from OS entropy
// only), --devices (batch-striped multi-GPU scan), --frames, --checkpoint (resumable scans).
#include <ctype.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../../include/vgen_hip.h"

namespace {

volatile int32_t g_stop = 0;

void on_sigint(int) {
    if (g_stop) _exit(130);   // second Ctrl-C exits (lib.rs:1088-1097)
    g_stop = 1;
}

struct Opts {
    std::string cmd, pattern, format = "p2pkh", output = "text", file, range, key, address, devices = "0", checkpoint, provider_table;
    bool has_pattern = false, ignore_case = false, quiet = false, json = false, no_gpu = false;
    uint64_t count = 1, repeat = 1, seed = 0;
    bool no_endo = false;
    bool random_keys = false;
    uint32_t batch = 1u << 20, frames = 12;   // twelve frames own twelve hardware queues (runtime.cpp)
    uint32_t table_bits_max = 0;              // vgen_scan_config.table_bits_max (0 = the context's memory policy decides)
    uint64_t mem_budget = 0;                  // vgen_params.device_mem_budget_bytes (0 = automatic)
    int puzzle = 0;
    long prefix_length = -1;   // -l / --prefix-length (provider patterns)
};

[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "Error: %s\n", msg.c_str());
    exit(1);
}

int format_id(const std::string &f) {
    if (f == "p2pkh") return VGEN_FMT_P2PKH;
    if (f == "p2wpkh") return VGEN_FMT_P2WPKH;
    if (f == "p2sh-p2wpkh" || f == "p2sh-p2-wpkh" || f == "p2shp2wpkh") return VGEN_FMT_P2SH_P2WPKH;
    if (f == "p2tr") return VGEN_FMT_P2TR;
    if (f == "ethereum") return VGEN_FMT_ETHEREUM;
    if (f == "p2pkh-uncompressed") return VGEN_FMT_P2PKH_UNCOMPRESSED;
    die("invalid value '" + f + "' for '--format' (p2pkh, p2wpkh, p2sh-p2wpkh, p2tr, ethereum)");
}

const char *format_display(int id) {   // Display for AddressFormat, src/address.rs:48-58
    switch (id) {
    case VGEN_FMT_P2PKH: return "P2PKH";
    case VGEN_FMT_P2PKH_UNCOMPRESSED: return "P2PKH (Uncompressed)";
    case VGEN_FMT_P2WPKH: return "P2WPKH";
    case VGEN_FMT_P2SH_P2WPKH: return "P2SH-P2WPKH";
    case VGEN_FMT_P2TR: return "P2TR";
    default: return "Ethereum";
    }
}

std::string format_duration(double secs) {   // lib.rs:1038-1052
    char b[64];
    if (secs < 1.0) snprintf(b, sizeof b, "%.0fms", secs * 1000.0);
    else if (secs < 60.0) snprintf(b, sizeof b, "%.1fs", secs);
    else if (secs < 3600.0) snprintf(b, sizeof b, "%.1fm", secs / 60.0);
    else if (secs < 86400.0) snprintf(b, sizeof b, "%.1fh", secs / 3600.0);
    else if (secs < 31536000.0) snprintf(b, sizeof b, "%.1fd", secs / 86400.0);
    else snprintf(b, sizeof b, "%.1fy", secs / 31536000.0);
    return b;
}

std::string with_commas(uint64_t n) {   // lib.rs:1075-1086
    std::string s = std::to_string(n), out;
    for (size_t i = 0; i < s.size(); i++) {
        if (i != 0 && (s.size() - i) % 3 == 0) out.push_back(',');
        out.push_back(s[i]);
    }
    return out;
}

std::string csv_escape(const std::string &f) {   // lib.rs:1058-1073
    if (f.find_first_of(",\"\n\r") == std::string::npos) return f;
    std::string out = "\"";
    for (char c : f) {
        if (c == '"') out.push_back('"');
        out.push_back(c);
    }
    return out + "\"";
}

std::string json_str(const std::string &s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') {
            o.push_back('\\');
            o.push_back((char)c);
        } else if (c < 0x20) {
            char b[8];
            snprintf(b, sizeof b, "\\u%04x", c);
            o += b;
        } else {
            o.push_back((char)c);
        }
    }
    return o + "\"";
}

std::string json_f64(double v) {
    char b[64];
    snprintf(b, sizeof b, "%.17g", v);
    double back = strtod(b, nullptr);
    for (int prec = 1; prec < 17; prec++) {   // shortest representation that round-trips
        char t[64];
        snprintf(t, sizeof t, "%.*g", prec, v);
        if (strtod(t, nullptr) == back) {
            snprintf(b, sizeof b, "%s", t);
            break;
        }
    }
    std::string s = b;
    if (s.find_first_of(".eE") == std::string::npos) s += ".0";
    return s;
}

bool parse_hex_key(const std::string &hex, uint8_t out[32]) {
    if (hex.empty() || hex.size() > 64) return false;
    memset(out, 0, 32);
    std::string h(64 - hex.size(), '0');
    h += hex;
    for (int i = 0; i < 32; i++) {
        unsigned v;
        if (sscanf(h.c_str() + 2 * i, "%2x", &v) != 1) return false;
        for (int j = 0; j < 2; j++) {
            char c = h[2 * i + j];
            if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'))) return false;
        }
        out[i] = (uint8_t)v;
    }
    return true;
}

// A mainnet WIF string (Base58Check of 0x80 | key [| 0x01]) taken apart WITHOUT its checksum being verified (the caller
// re-encodes the key and compares strings): the 32 key bytes, and whether the string claims a compressed public key.
bool parse_wif(const std::string &s, uint8_t key[32], bool *compressed) {
    static const char *alphabet = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
    if (s.size() < 50 || s.size() > 53) return false;
    std::vector<uint8_t> num;   // big-endian base-256 digits
    for (char ch : s) {
        const char *q = strchr(alphabet, ch);
        if (!q || !ch) return false;
        unsigned carry = (unsigned)(q - alphabet);
        for (size_t i = num.size(); i-- > 0;) {
            carry += 58u * num[i];
            num[i] = (uint8_t)carry;
            carry >>= 8;
        }
        while (carry) {
            num.insert(num.begin(), (uint8_t)carry);
            carry >>= 8;
        }
    }
    if ((num.size() != 37 && num.size() != 38) || num[0] != 0x80) return false;
    if (num.size() == 38 && num[33] != 0x01) return false;
    memcpy(key, num.data() + 1, 32);
    if (compressed) *compressed = num.size() == 38;
    return true;
}

void usage() {
    fprintf(stderr,
            "vgen-hip — MI355X scan engine for the vgen hot path\n\n"
            "  vgen-hip generate -p PATTERN [-f FORMAT] [-i] [-c COUNT] [-o text|json|jsonl|csv|minimal] [--file PATH]\n"
            "                    [--gpu-batch-size N] [--repeat N] [-q] [--seed S] [--devices 0,1,..|all] [--frames F]\n"
            "                    [--frames F]          (dispatches in flight, default 12; several searches on ONE device share it\n"
            "                                           evenly as long as their frames add up to twenty or fewer: 2 x 8, 3 x 6 ...)\n"
            "                    [--seed S]            (a reproducible search: base key / candidate stream derived from the 64-bit S —\n"
            "                                           for tests and benchmarks; keys found this way are only as secret as S)\n"
            "                    [--checkpoint FILE]   (resume an interrupted scan from FILE; written as the scan runs)\n"
            "                    [--mem-budget-gib G]  (device memory each context may hold; default: its frames, and no generator table\n"
            "                                           larger than half of the device's free memory)\n"
            "                    [--table-bits-max B]  (P2TR / --random-keys: the widest generator table a long scan may move to —\n"
            "                                           24: 11.8 GB, 27: 21.5 GB, 29: 138 GB (+12.5 %%); built behind the scan, never waited for)\n"
            "                    [--no-endo]           (unseeded searches, any format but P2TR, test six keys per curve\n"
            "                                           point — k, lambda k, lambda^2 k and their negations; this walks k0 + i only)\n"
            "                    [--random-keys]       (an independent random key per candidate, drawn on the device — the shape of\n"
            "                                           the reference's CPU path, src/scanner.rs:118-169 —, six keys per draw unless\n"
            "                                           --no-endo; ~3x slower than the walk)\n"
            "                    PATTERN may be a provider pattern boha:b1000:N [-l PREFIX_LENGTH] [--provider-table CSV]\n"
            "  vgen-hip range (--range START:END | --puzzle P) [-p PATTERN] [-f FORMAT] [-c COUNT (0 = whole range)] ...\n"
            "  vgen-hip estimate -p PATTERN [-f FORMAT] [-i]\n"
            "  vgen-hip verify -k WIF_OR_HEX [-a ADDRESS]\n"
            "  vgen-hip list-gpus [--json]\n");
}

Opts parse(int argc, char **argv) {
    Opts o;
    if (argc < 2) {
        usage();
        exit(2);
    }
    o.cmd = argv[1];
    if (o.cmd == "range") o.count = 1;
    // clap's spellings of one argument: --name=value, -nVALUE, and boolean shorts run together (-iq) — split up front.  Only
    // words in OPTION position are reinterpreted: the word after an option that takes a value is that value, verbatim, whatever
    // it starts with (`-p -abc` is the pattern "-abc", not three flags).
    static const char *const long_with_value[] = {"--pattern", "--format", "--count", "--output", "--file", "--gpu-batch-size", "--repeat",
                                                  "--seed", "--devices", "--frames", "--checkpoint", "--range", "--puzzle", "--key", "--address",
                                                  "--prefix-length", "--provider-table", "--threads", "--backend", "--cpu-batch-size", "--table-bits-max",
                                                  "--mem-budget-gib"};
    static const char short_with_value[] = "pfcorkaltT";
    std::vector<std::string> args;
    bool next_is_value = false;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (next_is_value) {
            args.push_back(a);
            next_is_value = false;
            continue;
        }
        const size_t eq = a.find('=');
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            if (eq != std::string::npos) {
                args.push_back(a.substr(0, eq));
                args.push_back(a.substr(eq + 1));
            } else {
                args.push_back(a);
                for (const char *name : long_with_value) next_is_value = next_is_value || a == name;
            }
        } else if (a.size() >= 2 && a[0] == '-' && a[1] != '-') {
            for (size_t k = 1; k < a.size(); k++) {
                const char ch = a[k];
                args.push_back(std::string("-") + ch);
                if (strchr(short_with_value, ch)) {   // takes a value: the rest of the word (after an optional '=') is it, else the next word
                    std::string rest = a.substr(k + 1);
                    if (!rest.empty() && rest[0] == '=') rest = rest.substr(1);
                    if (!rest.empty()) args.push_back(rest);
                    else next_is_value = true;
                    break;
                }
            }
        } else {
            args.push_back(a);
        }
    }
    for (size_t i = 0; i < args.size(); i++) {
        const std::string a = args[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= args.size()) die("a value is required for '" + a + "'");
            return args[++i];
        };
        if (a == "-p" || a == "--pattern") { o.pattern = val(); o.has_pattern = true; }
        else if (a == "-f" || a == "--format") o.format = val();
        else if (a == "-i" || a == "--ignore-case") o.ignore_case = true;
        else if (a == "-c" || a == "--count") o.count = strtoull(val().c_str(), nullptr, 10);
        else if (a == "-o" || a == "--output") o.output = val();
        else if (a == "--file") o.file = val();
        else if (a == "--gpu-batch-size") o.batch = (uint32_t)strtoul(val().c_str(), nullptr, 10);
        else if (a == "--repeat") o.repeat = strtoull(val().c_str(), nullptr, 10);
        else if (a == "-q" || a == "--quiet") o.quiet = true;
        else if (a == "--seed") o.seed = strtoull(val().c_str(), nullptr, 10);
        else if (a == "--devices") o.devices = val();
        else if (a == "--frames") o.frames = (uint32_t)strtoul(val().c_str(), nullptr, 10);
        else if (a == "--checkpoint") o.checkpoint = val();
        else if (a == "--table-bits-max") o.table_bits_max = (uint32_t)strtoul(val().c_str(), nullptr, 10);
        else if (a == "--mem-budget-gib") o.mem_budget = (uint64_t)(strtod(val().c_str(), nullptr) * 1073741824.0);
        else if (a == "-r" || a == "--range") o.range = val();
        else if (a == "--puzzle") o.puzzle = atoi(val().c_str());
        else if (a == "-k" || a == "--key") o.key = val();
        else if (a == "-a" || a == "--address") o.address = val();
        else if (a == "--json") o.json = true;
        else if (a == "--no-gpu") o.no_gpu = true;
        else if (a == "--no-endo") o.no_endo = true;
        else if (a == "--random-keys") o.random_keys = true;
        else if (a == "--no-tui" || a == "--tui") {}                                   // no TUI in this build
        else if (a == "-l" || a == "--prefix-length") o.prefix_length = strtol(val().c_str(), nullptr, 10);
        else if (a == "--provider-table") o.provider_table = val();
        else if (a == "-t" || a == "--threads" || a == "--backend" || a == "--cpu-batch-size") (void)val();   // accepted, not applicable
        else if (a == "-h" || a == "--help") { usage(); exit(0); }
        else die("unexpected argument '" + a + "'");
    }
    return o;
}

std::vector<int> parse_devices(const std::string &s) {
    int n = 0;
    if (vgen_device_count(&n) != VGEN_OK || n <= 0) die("no HIP device available (this build has no CPU backend)");
    std::vector<int> out;
    if (s == "all") {
        for (int i = 0; i < n; i++) out.push_back(i);
        return out;
    }
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t c = s.find(',', pos);
        std::string t = s.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        if (t.empty()) die("bad --devices list");
        int d = atoi(t.c_str());
        if (d < 0 || d >= n) die("device " + t + " out of range (" + std::to_string(n) + " device(s))");
        out.push_back(d);
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return out;
}

// resolve_pattern_and_format / the provider half of resolve_range_params (src/lib.rs:563-590,599-631):
// a "boha:..." pattern becomes the exact (or, with -l, prefix) pattern of the puzzle's address, selects the
// address format and, for `range` without an explicit range, supplies the key range.
struct Resolved {
    std::string pattern;
    bool from_provider = false, has_range = false;
    uint8_t start[32], end[32];
};

Resolved resolve_provider(Opts &o, bool for_range) {
    Resolved r;
    r.pattern = o.pattern;
    char addr[128];
    uint32_t fmt = 0;
    int32_t has_range = 0;
    const char *table = !o.provider_table.empty() ? o.provider_table.c_str() : getenv("VGEN_PROVIDER_TABLE");
    const int rc = vgen_provider_resolve(o.pattern.c_str(), table, addr, sizeof addr, &fmt, &has_range, r.start, r.end);
    if (rc < 0) die(vgen_last_error(nullptr));
    if (rc == 0) {
        if (o.prefix_length >= 0 && !for_range) fprintf(stderr, "Warning: --prefix-length is ignored for regex patterns\n");
        return r;
    }
    if (o.prefix_length == 0) die("--prefix-length must be at least 1 for provider patterns");
    char pat[300];
    vgen_provider_build_pattern(addr, o.prefix_length > 0 ? (uint32_t)o.prefix_length : 0, pat, sizeof pat);
    if (o.prefix_length > 0 || !for_range) fprintf(stderr, "Provider: %s → %s → pattern '%s'\n", o.pattern.c_str(), addr, pat);
    else fprintf(stderr, "Provider: %s → %s → exact match\n", o.pattern.c_str(), addr);
    static const char *names[] = {"p2pkh", "p2wpkh", "p2sh-p2wpkh", "p2tr", "p2pkh-uncompressed", "ethereum"};
    o.format = names[fmt];
    r.pattern = pat;
    r.from_provider = true;
    r.has_range = has_range != 0;
    return r;
}

// Warning for patterns that can never match (src/lib.rs:684-706).
void warn_impossible_pattern(const std::string &pattern, bool ignore_case, int fmt) {
    char bad[260];
    size_t n = 0;
    if (vgen_pattern_invalid_chars(pattern.c_str(), ignore_case, (uint32_t)fmt, bad, sizeof bad, &n) != VGEN_OK || n == 0) return;
    const char *cs = vgen_format_charset_name((uint32_t)fmt);
    fprintf(stderr, "Warning: Pattern contains characters not valid in %s addresses: '%s'\n", cs, bad);
    fprintf(stderr, "  %s alphabet excludes these characters - pattern will NEVER match!\n", cs);
    if (strcmp(cs, "Base58") == 0)
        fprintf(stderr, "  Base58 excludes: 0 (zero), O (uppercase o), I (uppercase i), l (lowercase L)\n");
    fprintf(stderr, "\n");
}

// `estimate` (src/lib.rs:345-375): difficulty heuristic over a measured rate.  The reference times its CPU
// generator (scanner.rs:333-346); here the rate is that of this pattern's scan on the device, over a
// bounded number of dispatches.
int run_estimate(Opts &o) {
    o.pattern = resolve_provider(o, false).pattern;
    const int fmt = format_id(o.format);
    vgen_filter *probe = nullptr;
    if (vgen_filter_compile(o.pattern.c_str(), o.ignore_case, (uint32_t)fmt, &probe) != VGEN_OK) die(vgen_last_error(nullptr));
    vgen_filter_free(probe);
    uint64_t difficulty = 0;
    vgen_pattern_difficulty(o.pattern.c_str(), o.ignore_case, (uint32_t)fmt, &difficulty);

    std::vector<int> devs = parse_devices(o.devices);
    vgen_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.device = devs[0];
    p.batch_size = o.batch;
    p.format = (uint32_t)fmt;
    p.frames = o.frames;
    vgen_ctx *c = nullptr;
    if (vgen_create(&p, &c) != VGEN_OK) die(std::string("GPU initialization failed: ") + vgen_last_error(nullptr));
    vgen_scan_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.format = (uint32_t)fmt;
    cfg.count = UINT64_MAX;
    cfg.case_insensitive = o.ignore_case;
    cfg.seed = o.seed;
    double rate = 0;
    for (int pass = 0; pass < 2; pass++) {   // first pass warms the device up
        cfg.max_batches = pass ? 96 : 12;
        vgen_scan_result res;
        if (vgen_scan(c, o.pattern.c_str(), &cfg, nullptr, nullptr, &g_stop, &res) != VGEN_OK) die(vgen_last_error(c));
        rate = res.elapsed_secs > 0 ? (double)res.operations / res.elapsed_secs : 0.0;
        vgen_scan_result_free(&res);
    }
    vgen_destroy(c);
    printf("Pattern: %s\nFormat: %s\nCase insensitive: %s\n\n", o.pattern.c_str(), format_display(fmt),
           o.ignore_case ? "true" : "false");
    printf("Estimated difficulty: 1 in %llu\n", (unsigned long long)difficulty);
    printf("Benchmark rate: %.0f addr/sec\n", rate);
    printf("Expected time: %s\n", format_duration(rate > 0 ? (double)difficulty / rate : 0.0).c_str());
    return 0;
}

// The reference's spinner line (src/lib.rs:783-805: "[elapsed] Checked N addresses", hidden when stderr is not a terminal and
// with --quiet): redrawn at most ten times a second from the scan's progress callback.
struct Progress {
    uint64_t base = 0;      // operations of the finished --repeat rounds
    double last = 0, t0 = 0;
    bool shown = false;
};
double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void progress_cb(uint64_t ops, void *user) {
    Progress *p = static_cast<Progress *>(user);
    const double t = now_s();
    if (t - p->last < 0.1) return;
    p->last = t;
    p->shown = true;
    const unsigned secs = (unsigned)(t - p->t0);
    fprintf(stderr, "\r[%02u:%02u:%02u] Checked %s addresses\x1b[K", secs / 3600, secs / 60 % 60, secs % 60, with_commas(p->base + ops).c_str());
    fflush(stderr);
}
void progress_clear(Progress &p) {
    if (p.shown) fprintf(stderr, "\r\x1b[K");
    p.shown = false;
}

int run_search(const Opts &o, const std::string &pattern, bool has_range, const uint8_t start[32], const uint8_t end[32]) {
    if (o.no_gpu) die("--no-gpu: this build has no CPU scan path (the MI355X engine is the only backend)");
    const int fmt = format_id(o.format);
    // surface pattern errors before touching the device (Pattern::new, pattern.rs:21-33)
    vgen_filter *probe = nullptr;
    if (vgen_filter_compile(pattern.c_str(), o.ignore_case, (uint32_t)fmt, &probe) != VGEN_OK) die(vgen_last_error(nullptr));
    vgen_filter_free(probe);
    warn_impossible_pattern(pattern, o.ignore_case, fmt);

    std::vector<int> devs = parse_devices(o.devices);
    std::vector<vgen_ctx *> ctxs;
    for (int d : devs) {
        vgen_params p;
        memset(&p, 0, sizeof p);
        p.struct_size = sizeof p;
        p.device = d;
        p.batch_size = o.batch;
        p.format = (uint32_t)fmt;
        p.frames = o.frames;
        p.device_mem_budget_bytes = o.mem_budget;
        // a vanity search proper — random base, no range, no seed, no checkpoint — may test any keys it likes:
        // six images per curve point (VGEN_FLAG_ENDO, +30 % keys per second); everything else walks k0 + i
        // (--random-keys: six keys per draw; seeds name candidate streams there, so they do not rule it out)
        if (!o.no_endo && fmt != 3 && (o.random_keys || (!has_range && !o.seed && o.checkpoint.empty()))) p.flags |= VGEN_FLAG_ENDO;
        vgen_ctx *c = nullptr;
        if (vgen_create(&p, &c) != VGEN_OK) die(std::string("GPU initialization failed: ") + vgen_last_error(nullptr));
        ctxs.push_back(c);
    }

    vgen_scan_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.format = (uint32_t)fmt;
    cfg.count = o.count == 0 ? UINT64_MAX : o.count;   // lib.rs:524
    cfg.case_insensitive = o.ignore_case;
    cfg.seed = o.seed;
    cfg.table_bits_max = o.table_bits_max;
    // every key a seeded search returns is a function of the 64-bit seed and a small counter: whoever learns the address can
    // replay the search (unseeded searches draw 256 / 192 bits from the OS, as the reference does: src/gpu.rs:936-945, scanner.rs:144)
    if (o.seed && !has_range && !o.quiet)
        fprintf(stderr, "Warning: --seed makes this search reproducible and its keys guessable (64 bits of secret): do not fund what it finds\n");
    if (has_range) {
        cfg.has_start = cfg.has_end = 1;
        memcpy(cfg.start, start, 32);
        memcpy(cfg.end, end, 32);
    }
    if (!o.checkpoint.empty()) cfg.checkpoint_path = o.checkpoint.c_str();   // resumable scan (not in the reference)
    if (o.random_keys) {
        if (has_range) die("--random-keys draws an independent key per candidate: no range");
        cfg.flags |= VGEN_SCAN_RANDOM_KEYS;
    }

    std::vector<vgen_generated> all;
    uint64_t total_ops = 0;
    double total_secs = 0;
    Progress prog;
    prog.t0 = now_s();
    const bool show_progress = !o.quiet && (isatty(2) || getenv("VGEN_PROGRESS") != nullptr);   // (VGEN_PROGRESS=1: also into a pipe or a log)
    for (uint64_t rep = 0; rep < (o.repeat ? o.repeat : 1) && !g_stop; rep++) {   // lib.rs:825-865
        vgen_scan_result res;
        prog.base = total_ops;
        int rc = ctxs.size() == 1 ? vgen_scan(ctxs[0], pattern.c_str(), &cfg, show_progress ? progress_cb : nullptr, &prog, &g_stop, &res)
                                  : vgen_scan_multi(ctxs.data(), (uint32_t)ctxs.size(), pattern.c_str(), &cfg, show_progress ? progress_cb : nullptr,
                                                    &prog, &g_stop, &res);
        progress_clear(prog);
        if (rc != VGEN_OK) die(vgen_last_error(ctxs[0]));
        // what the scan absorbed must not pass unseen: a device that failed (the others took its ranges over), a generator
        // table that could not be had (the scan went on with the small one, at a third of the rate)
        if (res.failed_shards > 0)
            fprintf(stderr, "Warning: %d of %zu devices failed during the scan; the remaining ones took their ranges over\n", res.failed_shards, ctxs.size());
        for (size_t i = 0; i < ctxs.size() && rep == 0; i++) {
            uint32_t bits = 0, wanted = 0;
            char note[256] = "";
            // (a note exists only for a table the context could NOT have: widths are not comparable as numbers — 27 signed bits beat 26)
            if (vgen_get_resources(ctxs[i], nullptr, &bits, &wanted, note, sizeof note) == VGEN_OK && bits && bits != wanted && note[0])
                fprintf(stderr, "Warning: device %zu: %s\n", i, note);
        }
        for (uint64_t i = 0; i < res.n_matches; i++) all.push_back(res.matches[i]);
        total_ops += res.operations;
        total_secs += res.elapsed_secs;
        vgen_scan_result_free(&res);
    }
    for (auto *c : ctxs) vgen_destroy(c);

    FILE *w = stdout;
    if (!o.file.empty() && !(w = fopen(o.file.c_str(), "w"))) die("Failed to create output file");
    const double rate = total_secs > 0 ? (double)total_ops / total_secs : 0.0;
    const std::string fmt_name = format_display(fmt);
    if (o.output == "csv" && !all.empty())
        fprintf(w, "address,wif,private_key_hex,format,pattern,operations,elapsed_secs,rate\n");
    for (size_t idx = 0; idx < all.size(); idx++) {
        const vgen_generated &g = all[idx];
        if (o.output == "text") {
            fprintf(w, "=== Match %zu of %zu ===\n", idx + 1, all.size());
            fprintf(w, "Pattern : %s\nFormat  : %s\nAddress : %s\nWIF     : %s\nHex     : %s\n", pattern.c_str(),
                    fmt_name.c_str(), g.address, g.wif, g.hex);
            if (!o.quiet) {
                fprintf(w, "Ops     : %s (%.0f/sec)\n", with_commas(total_ops).c_str(), rate);
                fprintf(w, "Time    : %s\n", format_duration(total_secs).c_str());
            }
            fprintf(w, "\n");
        } else if (o.output == "json" || o.output == "jsonl") {
            const bool pretty = o.output == "json";
            const char *nl = pretty ? "\n  " : "", *sp = pretty ? " " : "";
            fprintf(w, "{%s\"address\":%s%s,%s\"wif\":%s%s,%s\"private_key_hex\":%s%s,%s\"format\":%s%s,%s\"pattern\":%s%s,%s"
                       "\"operations\":%s%llu,%s\"elapsed_secs\":%s%s,%s\"rate\":%s%s%s}\n",
                    nl, sp, json_str(g.address).c_str(), nl, sp, json_str(g.wif).c_str(), nl, sp, json_str(g.hex).c_str(), nl,
                    sp, json_str(fmt_name).c_str(), nl, sp, json_str(pattern).c_str(), nl, sp,
                    (unsigned long long)total_ops, nl, sp, json_f64(total_secs).c_str(), nl, sp, json_f64(rate).c_str(),
                    pretty ? "\n" : "");
        } else if (o.output == "csv") {
            fprintf(w, "%s,%s,%s,%s,%s,%llu,%s,%s\n", csv_escape(g.address).c_str(), csv_escape(g.wif).c_str(),
                    csv_escape(g.hex).c_str(), csv_escape(fmt_name).c_str(), csv_escape(pattern).c_str(),
                    (unsigned long long)total_ops, json_f64(total_secs).c_str(), json_f64(rate).c_str());
        } else if (o.output == "minimal") {
            fprintf(w, "%s\n", g.wif);
        } else {
            die("invalid value '" + o.output + "' for '--output'");
        }
    }
    if (w != stdout) {
        fclose(w);
        if (!all.empty() && !o.quiet) fprintf(stderr, "Wrote %zu result(s) to %s\n", all.size(), o.file.c_str());
    }
    if (all.empty() && !o.quiet)
        fprintf(stderr, "No match found after %s operations (%s)\n", with_commas(total_ops).c_str(),
                format_duration(total_secs).c_str());
    return 0;
}

}  // namespace

const char *argv_pattern(int argc, char **argv) {
    for (int i = 2; i + 1 < argc; i++)
        if (!strcmp(argv[i], "-p") || !strcmp(argv[i], "--pattern")) return argv[i + 1];
    return "";
}

int main(int argc, char **argv) {

    Opts o = parse(argc, argv);
    signal(SIGINT, on_sigint);
    if (o.cmd == "generate") {
        if (!o.has_pattern) die("the following required arguments were not provided: --pattern <PATTERN>");
        uint8_t z[32] = {0};
        const Resolved r = resolve_provider(o, false);
        return run_search(o, r.pattern, false, z, z);
    }
    if (o.cmd == "estimate") {
        if (!o.has_pattern) die("the following required arguments were not provided: --pattern <PATTERN>");
        return run_estimate(o);
    }
    if (o.cmd == "range") {
        uint8_t start[32], end[32];
        Resolved pr;
        if (o.has_pattern) {
            pr = resolve_provider(o, true);
            o.pattern = pr.pattern;
        }
        if (pr.from_provider && !o.puzzle && o.range.empty()) {   // the provider's own key range, lib.rs:623-631
            if (!pr.has_range) die("Provider '" + std::string(argv_pattern(argc, argv)) + "' has no key range. Use --range or --puzzle to specify range.");
            memcpy(start, pr.start, 32);
            memcpy(end, pr.end, 32);
        } else if (o.puzzle) {   // lib.rs:643-649
            if (o.puzzle < 1 || o.puzzle > 160) die("Puzzle number must be between 1 and 160");
            memset(start, 0, 32);
            memset(end, 0, 32);
            const int sb = o.puzzle - 1;   // start = 2^(p-1), end = 2^p - 1
            start[31 - sb / 8] = (uint8_t)(1u << (sb % 8));
            for (int b = 0; b < o.puzzle; b++) end[31 - b / 8] |= (uint8_t)(1u << (b % 8));
        } else if (!o.range.empty()) {   // lib.rs:650-657
            size_t c = o.range.find(':');
            if (c == std::string::npos || o.range.find(':', c + 1) != std::string::npos) die("Range must be in format START:END");
            if (!parse_hex_key(o.range.substr(0, c), start)) die("Invalid start hex");
            if (!parse_hex_key(o.range.substr(c + 1), end)) die("Invalid end hex");
        } else {
            die("Either --range, --puzzle, or a provider pattern with key range must be specified");
        }
        bool start_zero = true;
        for (int i = 0; i < 32; i++) start_zero = start_zero && start[i] == 0;
        if (start_zero) start[31] = 1;   // key 0 is not a key: the CPU path skips it (scanner.rs:294-295)
        return run_search(o, o.has_pattern ? o.pattern : ".", true, start, end);   // default pattern, lib.rs:519
    }
    if (o.cmd == "list-gpus") {
        int n = 0;
        vgen_device_count(&n);
        if (o.json) printf("[");
        for (int i = 0; i < n; i++) {
            char name[256];
            vgen_device_name(i, name, sizeof name);
            if (o.json) printf("%s{\"index\":%d,\"name\":%s,\"backend\":\"hip\"}", i ? "," : "", i, json_str(name).c_str());
            else printf("[%d] %s (HIP)\n", i, name);
        }
        if (o.json) printf("]\n");
        if (!n && !o.json) printf("No GPU adapters found\n");
        return 0;
    }
    if (o.cmd == "verify") {
        // src/lib.rs:377-492: the key as WIF first, then as hex (an optional 0x in front); every address of the key; with
        // --address, MATCH when it is one of them (Bech32 in one case either way, Ethereum case-insensitively, raw 40 hex too)
        uint8_t k[32];
        bool is_wif = false;
        char addr[128], wif[128];
        if (parse_wif(o.key, k, nullptr)) {
            // the checksum: the key re-encoded in the form the string claims must be the string again
            bool compressed = false;
            (void)parse_wif(o.key, k, &compressed);
            if (vgen_derive(compressed ? VGEN_FMT_P2PKH : VGEN_FMT_P2PKH_UNCOMPRESSED, k, addr, sizeof addr, wif, sizeof wif) == VGEN_OK && o.key == wif)
                is_wif = true;
        }
        if (!is_wif) {
            std::string h = o.key;
            while (h.compare(0, 2, "0x") == 0) h = h.substr(2);   // trim_start_matches("0x")
            bool is_hex = !h.empty() && h.size() % 2 == 0;
            for (char c : h) is_hex = is_hex && isxdigit((unsigned char)c);
            if (!is_hex) die("Invalid key format (not WIF or hex)");
            if (h.size() != 64 || !parse_hex_key(h, k)) die("Hex key must be 32 bytes");
        }
        std::string a[6];
        const int fmts[6] = {VGEN_FMT_P2PKH, VGEN_FMT_P2PKH_UNCOMPRESSED, VGEN_FMT_P2WPKH, VGEN_FMT_P2SH_P2WPKH, VGEN_FMT_P2TR, VGEN_FMT_ETHEREUM};
        std::string wif_c, wif_u;
        for (int i = 0; i < 6; i++) {
            if (vgen_derive((uint32_t)fmts[i], k, addr, sizeof addr, wif, sizeof wif) != VGEN_OK) die("malformed or out-of-range secret key");
            a[i] = addr;
            if (fmts[i] == VGEN_FMT_P2PKH) wif_c = wif;
            if (fmts[i] == VGEN_FMT_P2PKH_UNCOMPRESSED) wif_u = wif;
        }
        std::string hex;
        for (int i = 0; i < 32; i++) {
            char b[3];
            snprintf(b, sizeof b, "%02x", k[i]);
            hex += b;
        }
        printf("Private key: %s\n", is_wif ? o.key.c_str() : wif_c.c_str());
        printf("WIF (uncompr.):     %s\n", wif_u.c_str());
        printf("Hex: %s\n\n", hex.c_str());
        printf("P2PKH address:      %s\n", a[0].c_str());
        printf("P2PKH (uncompr.):   %s\n", a[1].c_str());
        printf("P2WPKH address:     %s\n", a[2].c_str());
        printf("P2SH-P2WPKH addr:  %s\n", a[3].c_str());
        printf("P2TR address:       %s\n", a[4].c_str());
        printf("Ethereum address:   %s\n", a[5].c_str());
        if (!o.address.empty()) {
            auto lower = [](std::string t) {
                for (char &c : t) c = (char)tolower((unsigned char)c);
                return t;
            };
            const std::string &e = o.address;
            bool all_lower = true, all_upper = true, all_hex = true;
            for (char c : e) {
                if (isalpha((unsigned char)c)) {
                    all_lower = all_lower && islower((unsigned char)c);
                    all_upper = all_upper && isupper((unsigned char)c);
                }
                all_hex = all_hex && isxdigit((unsigned char)c);
            }
            const bool bech = e.size() >= 3 && lower(e.substr(0, 3)) == "bc1";
            const std::string norm = bech && (all_lower || all_upper) ? lower(e) : e;
            const std::string eth = norm.size() == 40 && all_hex ? "0x" + norm : norm;
            bool listed = false;
            for (const std::string &x : a) listed = listed || x == norm;
            if (listed) printf("\nMATCH!\n");
            else if (eth.size() >= 2 && lower(eth.substr(0, 2)) == "0x" && lower(a[5]) == lower(eth)) printf("\nMATCH! (Ethereum, case-insensitive)\n");
            else printf("\nMISMATCH! Expected: %s\n", e.c_str());
        }
        return 0;
    }
    usage();
    return 2;
}
