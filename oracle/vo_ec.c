/*
 * vo_ec.c — secp256k1 for the parity oracle (TEST INFRASTRUCTURE ONLY, see vgen_oracle.h).
 *
 * Restates what the reference obtains from libsecp256k1 through secp256k1-sys 0.10.1:
 *   SecretKey::from_slice            (src/address.rs:93;  src/gpu.rs:941,963)
 *   PublicKey::from_secret_key       (src/address.rs:96,122,127; src/gpu.rs:903-905)
 *   serialize / serialize_uncompressed
 *   x-only tweak-add for P2TR        (src/address.rs:136-140; src/gpu.rs:1288-1291)
 * Curve constants are cross-checked against the in-tree WGSL (src/shaders/field.wgsl:9-16 for p,
 * :346-347 for G) by tests/test_oracle_golden.py.
 *
 * Style: deliberately simple — 4x64-bit limbs, every field element kept fully reduced,
 * Fermat inversion by square-and-multiply, Jacobian coordinates with complete case handling.
 */
#include "vgen_oracle.h"
#include "vo_internal.h"

#include <pthread.h>
#include <string.h>

typedef unsigned __int128 u128;

/* p = 2^256 - 2^32 - 977 */
static const uint64_t P_LIMBS[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL,
                                    0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
#define P_C 0x1000003D1ULL /* 2^256 mod p */

/* group order n */
static const uint64_t N_LIMBS[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL,
                                    0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};

static const uint8_t G_X[32] = {0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62,
                                0x95, 0xCE, 0x87, 0x0B, 0x07, 0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE,
                                0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98};
static const uint8_t G_Y[32] = {0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB,
                                0xFC, 0x0E, 0x11, 0x08, 0xA8, 0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85,
                                0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};

/* ---- 256-bit helpers ------------------------------------------------------------------ */

static int u256_cmp(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}

static int u256_is_zero(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }

void vo_u256_from_be(const uint8_t in[32], uint64_t out[4]) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | in[(3 - i) * 8 + j];
        out[i] = w;
    }
}

void vo_u256_to_be(const uint64_t in[4], uint8_t out[32]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) out[(3 - i) * 8 + j] = (uint8_t)(in[i] >> (56 - 8 * j));
}

/* ---- field ---------------------------------------------------------------------------- */

void vo_fe_from_be(vo_fe *r, const uint8_t in[32]) { vo_u256_from_be(in, r->v); }
void vo_fe_to_be(const vo_fe *a, uint8_t out[32]) { vo_u256_to_be(a->v, out); }
int vo_fe_is_zero(const vo_fe *a) { return u256_is_zero(a->v); }
int vo_fe_is_odd(const vo_fe *a) { return (int)(a->v[0] & 1); }
int vo_fe_eq(const vo_fe *a, const vo_fe *b) { return u256_cmp(a->v, b->v) == 0; }

static void fe_cond_sub_p(uint64_t v[4], uint64_t carry) {
    /* value = carry*2^256 + v, known < 2p; bring into [0,p) */
    if (carry || u256_cmp(v, P_LIMBS) >= 0) {
        u128 b = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)v[i] - P_LIMBS[i] - (uint64_t)b;
            v[i] = (uint64_t)d;
            b = (d >> 64) & 1;
        }
    }
}

void vo_fe_add(vo_fe *r, const vo_fe *a, const vo_fe *b) {
    u128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        c += (u128)a->v[i] + b->v[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    fe_cond_sub_p(t, (uint64_t)c);
    memcpy(r->v, t, sizeof t);
}

void vo_fe_neg(vo_fe *r, const vo_fe *a) {
    if (u256_is_zero(a->v)) {
        memset(r->v, 0, sizeof r->v);
        return;
    }
    u128 b = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)P_LIMBS[i] - a->v[i] - (uint64_t)b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
    memcpy(r->v, t, sizeof t);
}

void vo_fe_sub(vo_fe *r, const vo_fe *a, const vo_fe *b) {
    vo_fe nb;
    vo_fe_neg(&nb, b);
    vo_fe_add(r, a, &nb);
}

void vo_fe_mul(vo_fe *r, const vo_fe *a, const vo_fe *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[i] * b->v[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    /* fold: value = lo + hi * 2^256 == lo + hi * P_C (mod p) */
    uint64_t s[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)t[i + 4] * P_C + t[i];
        s[i] = (uint64_t)c;
        c >>= 64;
    }
    s[4] = (uint64_t)c; /* < 2^34 */
    /* second fold of the (small) top limb */
    c = (u128)s[4] * P_C;
    uint64_t u[4];
    for (int i = 0; i < 4; i++) {
        c += s[i];
        u[i] = (uint64_t)c;
        c >>= 64;
    }
    /* c is 0 or 1; if 1 the low part is tiny, fold once more (adds P_C, cannot carry again) */
    if (c) {
        u128 d = (u128)u[0] + P_C;
        u[0] = (uint64_t)d;
        d >>= 64;
        for (int i = 1; i < 4 && d; i++) {
            d += u[i];
            u[i] = (uint64_t)d;
            d >>= 64;
        }
    }
    fe_cond_sub_p(u, 0);
    memcpy(r->v, u, sizeof u);
}

void vo_fe_sqr(vo_fe *r, const vo_fe *a) { vo_fe_mul(r, a, a); }

void vo_fe_inv(vo_fe *r, const vo_fe *a) {
    /* a^(p-2) by MSB-first square-and-multiply */
    uint64_t e[4] = {P_LIMBS[0] - 2, P_LIMBS[1], P_LIMBS[2], P_LIMBS[3]};
    vo_fe acc = {{1, 0, 0, 0}};
    for (int bit = 255; bit >= 0; bit--) {
        vo_fe_sqr(&acc, &acc);
        if ((e[bit >> 6] >> (bit & 63)) & 1) vo_fe_mul(&acc, &acc, a);
    }
    *r = acc;
}

/* ---- group ---------------------------------------------------------------------------- */

static void gej_set_inf(vo_gej *r) {
    memset(r, 0, sizeof *r);
    r->inf = 1;
}

void vo_gej_from_ge(vo_gej *r, const vo_ge *a) {
    if (a->inf) {
        gej_set_inf(r);
        return;
    }
    r->x = a->x;
    r->y = a->y;
    memset(&r->z, 0, sizeof r->z);
    r->z.v[0] = 1;
    r->inf = 0;
}

void vo_gej_double(vo_gej *r, const vo_gej *a) {
    if (a->inf || vo_fe_is_zero(&a->y)) {
        gej_set_inf(r);
        return;
    }
    vo_fe A, B, C, D, E, F, t, x3, y3, z3;
    vo_fe_sqr(&A, &a->x);
    vo_fe_sqr(&B, &a->y);
    vo_fe_sqr(&C, &B);
    vo_fe_add(&t, &a->x, &B);
    vo_fe_sqr(&t, &t);
    vo_fe_sub(&t, &t, &A);
    vo_fe_sub(&t, &t, &C);
    vo_fe_add(&D, &t, &t);
    vo_fe_add(&E, &A, &A);
    vo_fe_add(&E, &E, &A);
    vo_fe_sqr(&F, &E);
    vo_fe_sub(&x3, &F, &D);
    vo_fe_sub(&x3, &x3, &D);
    vo_fe_sub(&t, &D, &x3);
    vo_fe_mul(&y3, &E, &t);
    vo_fe_add(&C, &C, &C);
    vo_fe_add(&C, &C, &C);
    vo_fe_add(&C, &C, &C);
    vo_fe_sub(&y3, &y3, &C);
    vo_fe_mul(&z3, &a->y, &a->z);
    vo_fe_add(&z3, &z3, &z3);
    r->x = x3;
    r->y = y3;
    r->z = z3;
    r->inf = 0;
}

void vo_gej_add(vo_gej *r, const vo_gej *a, const vo_gej *b) {
    if (a->inf) {
        *r = *b;
        return;
    }
    if (b->inf) {
        *r = *a;
        return;
    }
    vo_fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t, hh, hhh, v, x3, y3, z3;
    vo_fe_sqr(&z1z1, &a->z);
    vo_fe_sqr(&z2z2, &b->z);
    vo_fe_mul(&u1, &a->x, &z2z2);
    vo_fe_mul(&u2, &b->x, &z1z1);
    vo_fe_mul(&t, &b->z, &z2z2);
    vo_fe_mul(&s1, &a->y, &t);
    vo_fe_mul(&t, &a->z, &z1z1);
    vo_fe_mul(&s2, &b->y, &t);
    vo_fe_sub(&h, &u2, &u1);
    vo_fe_sub(&rr, &s2, &s1);
    if (vo_fe_is_zero(&h)) {
        if (vo_fe_is_zero(&rr))
            vo_gej_double(r, a);
        else
            gej_set_inf(r);
        return;
    }
    vo_fe_sqr(&hh, &h);
    vo_fe_mul(&hhh, &hh, &h);
    vo_fe_mul(&v, &u1, &hh);
    vo_fe_sqr(&x3, &rr);
    vo_fe_sub(&x3, &x3, &hhh);
    vo_fe_sub(&x3, &x3, &v);
    vo_fe_sub(&x3, &x3, &v);
    vo_fe_sub(&t, &v, &x3);
    vo_fe_mul(&y3, &rr, &t);
    vo_fe_mul(&t, &s1, &hhh);
    vo_fe_sub(&y3, &y3, &t);
    vo_fe_mul(&z3, &a->z, &b->z);
    vo_fe_mul(&z3, &z3, &h);
    r->x = x3;
    r->y = y3;
    r->z = z3;
    r->inf = 0;
}

void vo_gej_add_ge(vo_gej *r, const vo_gej *a, const vo_ge *b) {
    vo_gej bj;
    vo_gej_from_ge(&bj, b);
    vo_gej_add(r, a, &bj);
}

void vo_ge_from_gej(vo_ge *r, const vo_gej *a) {
    if (a->inf) {
        memset(r, 0, sizeof *r);
        r->inf = 1;
        return;
    }
    vo_fe zi, zi2, zi3;
    vo_fe_inv(&zi, &a->z);
    vo_fe_sqr(&zi2, &zi);
    vo_fe_mul(&zi3, &zi2, &zi);
    vo_fe_mul(&r->x, &a->x, &zi2);
    vo_fe_mul(&r->y, &a->y, &zi3);
    r->inf = 0;
}

void vo_ge_generator(vo_ge *g) {
    vo_fe_from_be(&g->x, G_X);
    vo_fe_from_be(&g->y, G_Y);
    g->inf = 0;
}

/* ---- fixed-base table: 64 windows x 15 non-zero 4-bit digits -------------------------------
 * (the windowed fixed-base structure libsecp256k1's ecmult_gen uses; here without blinding) */

static vo_ge g_table[64][16];
static pthread_once_t g_table_once = PTHREAD_ONCE_INIT;

static void build_table(void) {
    vo_ge g;
    vo_ge_generator(&g);
    vo_gej base;
    vo_gej_from_ge(&base, &g);
    for (int w = 0; w < 64; w++) {
        vo_gej acc;
        gej_set_inf(&acc);
        memset(&g_table[w][0], 0, sizeof(vo_ge));
        g_table[w][0].inf = 1;
        for (int d = 1; d < 16; d++) {
            vo_gej_add(&acc, &acc, &base);
            vo_ge_from_gej(&g_table[w][d], &acc);
        }
        /* base <- 16 * base */
        for (int k = 0; k < 4; k++) vo_gej_double(&base, &base);
    }
}

void vo_ecmult_gen(vo_gej *r, const uint64_t k[4]) {
    pthread_once(&g_table_once, build_table);
    gej_set_inf(r);
    for (int w = 0; w < 64; w++) {
        unsigned d = (unsigned)((k[w >> 4] >> ((w & 15) * 4)) & 15);
        if (d) vo_gej_add_ge(r, r, &g_table[w][d]);
    }
}

void vo_ecmult_naive(vo_gej *r, const uint64_t k[4]) {
    vo_ge g;
    vo_ge_generator(&g);
    gej_set_inf(r);
    for (int bit = 255; bit >= 0; bit--) {
        vo_gej_double(r, r);
        if ((k[bit >> 6] >> (bit & 63)) & 1) vo_gej_add_ge(r, r, &g);
    }
}

/* ---- public API ------------------------------------------------------------------------- */

int vo_scalar_valid(const uint64_t k[4]) { return !u256_is_zero(k) && u256_cmp(k, N_LIMBS) < 0; }

int vo_key_valid(const uint8_t key_be[32]) {
    uint64_t k[4];
    vo_u256_from_be(key_be, k);
    return vo_scalar_valid(k);
}

static int pub_from_key(const uint8_t key_be[32], uint8_t pub65[65], int naive) {
    uint64_t k[4];
    vo_u256_from_be(key_be, k);
    if (!vo_scalar_valid(k)) return 0;
    vo_gej pj;
    if (naive)
        vo_ecmult_naive(&pj, k);
    else
        vo_ecmult_gen(&pj, k);
    vo_ge pa;
    vo_ge_from_gej(&pa, &pj);
    if (pa.inf) return 0;
    pub65[0] = 0x04;
    vo_fe_to_be(&pa.x, pub65 + 1);
    vo_fe_to_be(&pa.y, pub65 + 33);
    return 1;
}

int vo_pubkey(const uint8_t key_be[32], uint8_t pub65[65]) { return pub_from_key(key_be, pub65, 0); }
int vo_pubkey_naive(const uint8_t key_be[32], uint8_t pub65[65]) { return pub_from_key(key_be, pub65, 1); }

int vo_key_add_u64(const uint8_t key_be[32], uint64_t amount, uint8_t out_be[32]) {
    /* plain 256-bit add, as increment_key does byte-wise (src/gpu.rs:951-961) */
    uint64_t k[4];
    vo_u256_from_be(key_be, k);
    u128 c = amount;
    for (int i = 0; i < 4; i++) {
        c += k[i];
        k[i] = (uint64_t)c;
        c >>= 64;
    }
    vo_u256_to_be(k, out_be);
    return (int)c;
}

int vo_lift_x(const uint8_t x_be[32], uint8_t pub65[65]) {
    /* BIP-340 lift_x: y = (x^3 + 7)^((p+1)/4), even root; fails if x is not on the curve */
    vo_fe x, y2, y, t;
    vo_fe_from_be(&x, x_be);
    vo_fe_sqr(&t, &x);
    vo_fe_mul(&y2, &t, &x);
    vo_fe seven = {{7, 0, 0, 0}};
    vo_fe_add(&y2, &y2, &seven);
    /* e = (p + 1) / 4 */
    uint64_t e[4] = {(P_LIMBS[0] + 1) >> 2 | (P_LIMBS[1] << 62), (P_LIMBS[1] >> 2) | (P_LIMBS[2] << 62),
                     (P_LIMBS[2] >> 2) | (P_LIMBS[3] << 62), P_LIMBS[3] >> 2};
    vo_fe acc = {{1, 0, 0, 0}};
    for (int bit = 255; bit >= 0; bit--) {
        vo_fe_sqr(&acc, &acc);
        if ((e[bit >> 6] >> (bit & 63)) & 1) vo_fe_mul(&acc, &acc, &y2);
    }
    y = acc;
    vo_fe_sqr(&t, &y);
    if (!vo_fe_eq(&t, &y2)) return 0;
    if (vo_fe_is_odd(&y)) vo_fe_neg(&y, &y);
    pub65[0] = 0x04;
    vo_fe_to_be(&x, pub65 + 1);
    vo_fe_to_be(&y, pub65 + 33);
    return 1;
}

int vo_taproot_output_key(const uint8_t pub65[65], uint8_t out_x[32]) {
    /* BIP-341: P = lift_x(x(internal));  t = tagged_hash("TapTweak", x(P));  Q = P + t*G */
    vo_ge p;
    vo_fe_from_be(&p.x, pub65 + 1);
    vo_fe_from_be(&p.y, pub65 + 33);
    p.inf = 0;
    if (vo_fe_is_odd(&p.y)) vo_fe_neg(&p.y, &p.y);

    uint8_t tag[32], buf[96], t_be[32];
    vo_sha256((const uint8_t *)"TapTweak", 8, tag);
    memcpy(buf, tag, 32);
    memcpy(buf + 32, tag, 32);
    memcpy(buf + 64, pub65 + 1, 32);
    vo_sha256(buf, 96, t_be);

    uint64_t t[4];
    vo_u256_from_be(t_be, t);
    if (u256_cmp(t, N_LIMBS) >= 0) return 0;
    vo_gej tj, qj;
    vo_ecmult_gen(&tj, t);
    vo_gej_add_ge(&qj, &tj, &p);
    vo_ge q;
    vo_ge_from_gej(&q, &qj);
    if (q.inf) return 0;
    vo_fe_to_be(&q.x, out_x);
    return 1;
}
