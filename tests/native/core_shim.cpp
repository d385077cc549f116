// core_shim.cpp — exposes the single-source core headers (vgen_amd/csrc/core/*.h), compiled for the
// HOST by g++, to the CPU test-suite through ctypes.  The same headers are what hipcc compiles into
// the kernels, so arithmetic bugs (limb overflow, magnitude violations) are caught without a GPU.
#include "../../vgen_amd/csrc/core/fe.h"

using namespace vg;

extern "C" {
void core_fe_mul(const u32 *a, const u32 *b, u32 *r) {
    fe x, y, z;
    for (int i = 0; i < 9; i++) { x.n[i] = a[i]; y.n[i] = b[i]; }
    fe_mul(z, x, y);
    for (int i = 0; i < 9; i++) r[i] = z.n[i];
}
void core_fe_sqr(const u32 *a, u32 *r) {
    fe x, z;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_sqr(z, x);
    for (int i = 0; i < 9; i++) r[i] = z.n[i];
}
void core_fe_normalize(const u32 *a, u32 *r, int weak) {
    fe x;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    if (weak) fe_normalize_weak(x); else fe_normalize(x);
    for (int i = 0; i < 9; i++) r[i] = x.n[i];
}
void core_fe_neg(const u32 *a, u32 m, u32 *r) {
    fe x, z;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_neg(z, x, m);
    for (int i = 0; i < 9; i++) r[i] = z.n[i];
}
void core_fe_inv(const u32 *a, u32 *r) {
    fe x, z;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_inv(z, x);
    for (int i = 0; i < 9; i++) r[i] = z.n[i];
}
void core_fe_inv_fermat(const u32 *a, u32 *r) {
    fe x, z;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_inv_fermat(z, x);
    for (int i = 0; i < 9; i++) r[i] = z.n[i];
}
// One batch of 29 divsteps on given low words (unit test of the matrix against the textbook recurrence).
// the inversion's final lift: signed limbs of d in (-2p, p), the sign word of f -> canonical sign(f) * d mod p
void core_fe_divsteps_lift(const int *d, int f_top, u32 *r) {
    fe_sgn ds;
    for (int i = 0; i < 9; i++) ds.n[i] = d[i];
    fe o;
    fe_divsteps_lift_(o, ds, f_top);
    for (int i = 0; i < 9; i++) r[i] = o.n[i];
}

int core_fe_divsteps29(int zeta, u32 f, u32 g, int *t) {
    int32_t tt[4];
    const int32_t z = fe_divsteps29_(zeta, f, g, tt);
    for (int i = 0; i < 4; i++) t[i] = tt[i];
    return z;
}
void core_fe_words(const u32 *a, u32 *w, u32 *back) {
    fe x, y;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_to_words(x, w);
    fe_from_words(y, w);
    for (int i = 0; i < 9; i++) back[i] = y.n[i];
}
}

#include "../../vgen_amd/csrc/core/hash.h"

extern "C" {
// x, y: 32 big-endian bytes each
static void be32_to_words(const unsigned char *be, u32 w[8]) {
    for (int i = 0; i < 8; i++)
        w[i] = ((u32)be[28 - 4 * i] << 24) | ((u32)be[29 - 4 * i] << 16) | ((u32)be[30 - 4 * i] << 8) | be[31 - 4 * i];
}
void core_hash160_pub33(u32 prefix, const unsigned char *x_be, unsigned char *out20) {
    u32 xw[8], sha[8], h[5];
    be32_to_words(x_be, xw);
    sha256_pub33(prefix, xw, sha);
    ripemd160_of_sha(sha, h);
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) out20[4 * i + j] = (unsigned char)(h[i] >> (8 * j));
}
void core_hash160_pub65(const unsigned char *x_be, const unsigned char *y_be, unsigned char *out20) {
    u32 xw[8], yw[8], sha[8], h[5];
    be32_to_words(x_be, xw);
    be32_to_words(y_be, yw);
    sha256_pub65(xw, yw, sha);
    ripemd160_of_sha(sha, h);
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) out20[4 * i + j] = (unsigned char)(h[i] >> (8 * j));
}
void core_hash160_script22(const unsigned char *h160, unsigned char *out20) {
    u32 hin[5], sha[8], h[5];
    for (int i = 0; i < 5; i++) hin[i] = (u32)h160[4 * i] | ((u32)h160[4 * i + 1] << 8) | ((u32)h160[4 * i + 2] << 16) | ((u32)h160[4 * i + 3] << 24);
    sha256_script22(hin, sha);
    ripemd160_of_sha(sha, h);
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) out20[4 * i + j] = (unsigned char)(h[i] >> (8 * j));
}
void core_keccak_addr(const unsigned char *x_be, const unsigned char *y_be, unsigned char *out20) {
    u32 xw[8], yw[8], h[5];
    be32_to_words(x_be, xw);
    be32_to_words(y_be, yw);
    keccak256_pub64_addr(xw, yw, h);
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) out20[4 * i + j] = (unsigned char)(h[i] >> (8 * j));
}
}

#include "../../vgen_amd/csrc/host/host_ec.h"

extern "C" {
// key (32 BE bytes) -> x||y (64 BE bytes); returns 0 for k == 0
int core_mul_gen(const unsigned char *key_be, unsigned char *xy) {
    Scalar k;
    scalar_from_be(k, key_be);
    ge p;
    if (!host_ec_mul_gen(k, p)) return 0;
    u32 w[8];
    fe_to_words(p.x, w);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[4 * (7 - i) + j] = (unsigned char)(w[i] >> (24 - 8 * j));
    fe_to_words(p.y, w);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[32 + 4 * (7 - i) + j] = (unsigned char)(w[i] >> (24 - 8 * j));
    return 1;
}
// host_seq_points on one persistent cache: call k asks for the S points of base key kb_be[k] (32 bytes each);
// xy receives S * 64 bytes per call (x||y big-endian).  Returns the number of successful calls.
int core_seq_points_walk(const unsigned char *kb_be, int calls, unsigned S, unsigned char *xy) {
    SeqBaseCache cache;
    int ok = 0;
    for (int k = 0; k < calls; k++) {
        Scalar kb;
        scalar_from_be(kb, kb_be + 32 * k);
        ge pts[32];
        if (!host_seq_points(cache, kb, S, pts)) continue;
        ok++;
        for (unsigned e = 0; e < S; e++) {
            unsigned char *o = xy + ((size_t)k * S + e) * 64;
            u32 w[8];
            fe_to_words(pts[e].x, w);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) o[4 * (7 - i) + j] = (unsigned char)(w[i] >> (24 - 8 * j));
            fe_to_words(pts[e].y, w);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) o[32 + 4 * (7 - i) + j] = (unsigned char)(w[i] >> (24 - 8 * j));
        }
    }
    return ok;
}
// stride table entry check: out = x||y of entries
void core_stride_table(uint64_t first, uint64_t step, uint32_t count, unsigned char *xy) {
    std::vector<ge> t;
    host_build_stride_table(first, step, count, t);
    for (uint32_t e = 0; e < count; e++) {
        u32 w[8];
        fe_to_words(t[e].x, w);
        for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[64 * e + 4 * (7 - i) + j] = (unsigned char)(w[i] >> (24 - 8 * j));
        fe_to_words(t[e].y, w);
        for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[64 * e + 32 + 4 * (7 - i) + j] = (unsigned char)(w[i] >> (24 - 8 * j));
    }
}
}

#include <string.h>
#include <string>
#include "../../vgen_amd/csrc/core/filter_eval.h"
#include "../../vgen_amd/csrc/host/encode.h"
#include "../../vgen_amd/csrc/host/filter.h"

extern "C" {
// Compiles pattern for `format`; evaluates the DEVICE prefilter program (the same filter_eval the
// kernel runs) and the exact DFA on the encoded address of `payload`.
// returns: bit0 = device prefilter hit, bit1 = exact match, or -1 on pattern error. kind_out = device kind.
// host_sha256 of an arbitrary message, by the portable block function (0) or with the SHA extensions allowed (1)
void core_host_sha256(const unsigned char *msg, unsigned long len, unsigned char *out32, int allow_sha_ni) {
    host_sha256_with(msg, len, out32, allow_sha_ni != 0);
}

int core_filter_check(const char *pattern, int ci, unsigned format, const unsigned char *payloads, int n,
                      unsigned char *out_flags, int *kind_out, double *sel_out) {
    vgen_filter f;
    std::string err;
    if (!filter_compile(pattern, ci != 0, format, f, err)) return -1;
    *kind_out = (int)f.dev.kind;
    *sel_out = f.selectivity;
    for (int i = 0; i < n; i++) {
        const int pb = format == 3 ? 32 : 20;     // P2TR carries the 32-byte x-only output key
        u32 pl[8];
        for (int w = 0; w < pb / 4; w++) {
            const unsigned char *p = payloads + pb * i + 4 * w;
            pl[w] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
        }
        int dev = (format == 3 ? filter_eval_n<8>(&f.dev, pl) : filter_eval(&f.dev, pl)) ? 1 : 0;
        std::string addr = address_from_payload(format, payloads + pb * i);
        int exact = f.dfa.is_match(addr) ? 1 : 0;
        out_flags[i] = (unsigned char)(dev | (exact << 1));
    }
    return 0;
}
// Number of device tests the filter compiler produced (ranges / masked tests), or -1 on pattern error.
int core_filter_ntests(const char *pattern, int ci, unsigned format) {
    vgen_filter f;
    std::string err;
    if (!filter_compile(pattern, ci != 0, format, f, err)) return -1;
    return (int)f.dev.count;
}
int core_regex_match(const char *pattern, int ci, const char *text) {
    Dfa d;
    std::string err;
    if (!regex_compile(pattern, ci != 0, d, err)) return -1;
    return d.is_match(text) ? 1 : 0;
}
int core_address(unsigned format, const unsigned char *payload, char *out) {
    std::string s = address_from_payload(format, payload);
    strcpy(out, s.c_str());
    return (int)s.size();
}
}

extern "C" {
// Fixed-window accumulation exactly as keys_scan_kernel does it (unsigned 4-bit windows, low to high,
// gej_add_ge_nz without special cases) -> x||y big-endian; returns 0 for k == 0.
int core_mul_windows_nz(const unsigned char *key_be, unsigned char *xy) {
    Scalar k;
    scalar_from_be(k, key_be);
    ge g;
    ge_generator(g);
    gej base;
    gej_from_ge(base, g);
    gej acc;
    gej_set_infinity(acc);
    for (int w = 0; w < 64; w++) {
        unsigned d = (k.w[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (d) {
            // d * 16^w * G by repeated generic addition (reference for the table entry)
            gej e;
            gej_set_infinity(e);
            ge ba;
            ge_from_gej(ba, base);
            for (unsigned t = 0; t < d; t++) gej_add_ge(e, e, ba);
            ge ea;
            ge_from_gej(ea, e);
            if (acc.inf) gej_from_ge(acc, ea);
            else gej_add_ge_nz(acc, acc, ea);
        }
        for (int t = 0; t < 4; t++) gej_double(base, base);
    }
    ge p;
    if (!ge_from_gej(p, acc)) return 0;
    u32 wv[8];
    fe_to_words(p.x, wv);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[4 * (7 - i) + j] = (unsigned char)(wv[i] >> (24 - 8 * j));
    fe_to_words(p.y, wv);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[32 + 4 * (7 - i) + j] = (unsigned char)(wv[i] >> (24 - 8 * j));
    return 1;
}
}

extern "C" {
void core_fe_mul_add(const u32 *a, const u32 *b, const u32 *c, u32 *r, int square) {
    fe x, y, z, w;
    for (int i = 0; i < 9; i++) { x.n[i] = a[i]; y.n[i] = b[i]; z.n[i] = c[i]; }
    if (square) fe_sqr_add(w, x, z); else fe_mul_add(w, x, y, z);
    for (int i = 0; i < 9; i++) r[i] = w.n[i];
}
u32 core_fe_parity_weak(const u32 *a) {
    fe x;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    return fe_parity_weak(x);
}
void core_fe_canonicalize_product(const u32 *a, u32 *r) {
    fe x;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_canonicalize_product(x);
    for (int i = 0; i < 9; i++) r[i] = x.n[i];
}
void core_fe_canonicalize(const u32 *a, u32 *r) {
    fe x;
    for (int i = 0; i < 9; i++) x.n[i] = a[i];
    fe_canonicalize(x);
    for (int i = 0; i < 9; i++) r[i] = x.n[i];
}
}

#include "../../vgen_amd/csrc/core/dfa_eval.h"

extern "C" {
// Runs the DEVICE full-match algorithm (dfa_match_payload, the code the kernel runs) and the exact DFA on
// the host-encoded address for every payload.  out_flags: bit0 device-dfa result, bit1 exact result.
// returns the device kind the filter compiler chose, or -1 on pattern error.
int core_dfa_check(const char *pattern, int ci, unsigned format, const unsigned char *payloads, int n,
                   unsigned char *out_flags) {
    vgen_filter f;
    std::string err;
    if (!filter_compile(pattern, ci != 0, format, f, err)) return -1;
    if (f.dev.kind != DEVF_DFA) return (int)f.dev.kind;
    for (int i = 0; i < n; i++) {
        const int pb = format == 3 ? 32 : 20;
        u32 pl[8];
        for (int w = 0; w < pb / 4; w++) {
            const unsigned char *p = payloads + pb * i + 4 * w;
            pl[w] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
        }
        int dev = (format == 3 ? dfa_match_payload_n<8>(f.dfa_blob.data(), 3, pl)
                               : dfa_match_payload(f.dfa_blob.data(), (int)format, pl)) ? 1 : 0;
        std::string addr = address_from_payload(format, payloads + pb * i);
        int exact = f.dfa.is_match(addr) ? 1 : 0;
        out_flags[i] = (unsigned char)(dev | (exact << 1));
    }
    return (int)DEVF_DFA;
}
}

#include "../../vgen_amd/csrc/core/taproot.h"

extern "C" {
// Device algorithm for the P2TR output key, on the host: key -> k*G via the 4-bit AND the 8-bit fixed-window
// multiplications (must agree), then taproot_output_x.  out: 32 bytes big-endian x(Q).  0 if invalid,
// -1 if the two multiplications disagree.
int core_taproot_from_key(const unsigned char *key_be, unsigned char *out32) {
    static std::vector<uint32_t> tab, tab8;
    if (tab.empty()) {
        host_gen_table_limbs(tab);
        host_gen_table8_limbs(tab8);
    }
    Scalar k;
    scalar_from_be(k, key_be);
    if (!scalar_is_valid(k)) return 0;
    gej pj, pj8;
    ec_mul_gen_windows(pj, k.w, tab.data());
    ec_mul_gen_w8(pj8, k.w, tab8.data());
    ge p, p8;
    if (!ge_from_gej(p, pj) || !ge_from_gej(p8, pj8)) return 0;
    if (!fe_equal_canonical(p.x, p8.x) || !fe_equal_canonical(p.y, p8.y)) return -1;
    u32 xw[8];
    if (!taproot_output_x(p.x, p.y, tab8.data(), xw)) return 0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) out32[4 * (7 - i) + j] = (unsigned char)(xw[i] >> (24 - 8 * j));
    return 1;
}
}

extern "C" {
// ec_mul_gen_w16 (core/ec.h, the device algorithm of the KEYS / P2TR paths) on the host: the 67 MB table is
// allocated untouched and only the sixteen entries this key walks are filled in (d * 2^(16 w) * G by the generic host
// multiplication); result x||y big-endian.  Returns 0 for k == 0.
int core_mul_w16(const unsigned char *key_be, unsigned char *xy) {
    static uint32_t *tab = (uint32_t *)calloc(EC_TABLE16_WORDS, sizeof(uint32_t));
    Scalar k;
    scalar_from_be(k, key_be);
    if (!scalar_is_valid(k)) return 0;
    for (int w = 0; w < 16; w++) {
        const uint32_t d = (k.w[w >> 1] >> ((w & 1) * 16)) & 0xFFFFu;
        if (!d) continue;
        Scalar e;
        for (int i = 0; i < 8; i++) e.w[i] = 0;
        e.w[w >> 1] = d << ((w & 1) * 16);
        ge p;
        if (!host_ec_mul_gen(e, p)) return 0;
        uint32_t *o = tab + ((size_t)w * 65535u + (d - 1)) * 16;
        fe_to_words(p.x, o);
        fe_to_words(p.y, o + 8);
    }
    gej acc;
    ec_mul_gen_w16(acc, k.w, tab);
    ge r;
    if (!ge_from_gej(r, acc)) return 0;
    u32 wv[8];
    fe_to_words(r.x, wv);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[4 * (7 - i) + j] = (unsigned char)(wv[i] >> (24 - 8 * j));
    fe_to_words(r.y, wv);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[32 + 4 * (7 - i) + j] = (unsigned char)(wv[i] >> (24 - 8 * j));
    return 1;
}
}

#include <sys/mman.h>
extern "C" {
// ec_mul_gen_signed<ST> (core/ec.h: signed windows of 25 / 27 / 29 bits, the device algorithm behind VGEN_GTAB_BITS = 25 | 27 | 29) on
// the host: the table (5.9 / 21.5 / 138 GB) is mapped without backing store and only the entries this key walks are filled
// in — m * 2^(ST w) * G by the generic host multiplication, the top window's 2^256 taken mod n as the device builder does.
// result x||y big-endian.  Returns 0 for an invalid key, -1 when the table cannot be mapped.
int core_mul_signed(int st, const unsigned char *key_be, unsigned char *xy) {
    static uint32_t *tabs[32] = {nullptr};
    if (st != 25 && st != 27 && st != 29) return -1;
    if (!tabs[st]) {
        void *p = mmap(nullptr, (size_t)ec_table_words((u32)st) * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return -1;
        tabs[st] = (uint32_t *)p;
    }
    uint32_t *tab = tabs[st];
    Scalar k;
    scalar_from_be(k, key_be);
    if (!scalar_is_valid(k)) return 0;
    const u32 nw = ec_signed_windows((u32)st);
    u32 carry = 0;
    for (u32 w = 0; w < nw; w++) {
        const u32 v = ec_wide_digit(k.w, w, (u32)st) + carry;
        const bool neg = v > (1u << (st - 1));
        const u32 m = neg ? (1u << st) - v : v;
        carry = neg ? 1u : 0u;
        if (!m) continue;
        // scalar m * 2^(st w), as 9 words; exactly 2^256 -> 2^256 - n
        const u32 bit = (u32)st * w;
        uint32_t e9[10] = {0};
        const unsigned long long lo = (unsigned long long)m << (bit & 31u);
        e9[bit >> 5] = (uint32_t)lo;
        e9[(bit >> 5) + 1] = (uint32_t)(lo >> 32);
        Scalar e;
        for (int i = 0; i < 8; i++) e.w[i] = e9[i];
        if (e9[8]) {
            if (e9[8] != 1) return -1;
            for (int i = 0; i < 8; i++) if (e9[i]) return -1;
            static const uint32_t r[8] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u, 0, 0, 0};
            for (int i = 0; i < 8; i++) e.w[i] = r[i];
        }
        ge p;
        if (!host_ec_mul_gen(e, p)) return 0;
        uint32_t *o = tab + ((size_t)w * ec_signed_per((u32)st) + (m - 1)) * 16;
        fe_to_words(p.x, o);
        fe_to_words(p.y, o + 8);
    }
    if (carry) return -1;   // cannot happen for k < 2^256
    gej acc;
    if (st == 25) ec_mul_gen_signed<25>(acc, k.w, tab);
    else if (st == 27) ec_mul_gen_signed<27>(acc, k.w, tab);
    else ec_mul_gen_signed<29>(acc, k.w, tab);
    ge r;
    if (!ge_from_gej(r, acc)) return 0;
    u32 wv[8];
    fe_to_words(r.x, wv);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[4 * (7 - i) + j] = (unsigned char)(wv[i] >> (24 - 8 * j));
    fe_to_words(r.y, wv);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) xy[32 + 4 * (7 - i) + j] = (unsigned char)(wv[i] >> (24 - 8 * j));
    return 1;
}
}

