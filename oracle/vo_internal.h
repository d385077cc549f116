/* vo_internal.h — types shared between the oracle's translation units (TEST INFRASTRUCTURE ONLY). */
#ifndef VO_INTERNAL_H
#define VO_INTERNAL_H

#include <stdint.h>

typedef struct {
    uint64_t v[4]; /* little-endian limbs, always fully reduced mod p */
} vo_fe;

typedef struct {
    vo_fe x, y;
    int inf;
} vo_ge;

typedef struct {
    vo_fe x, y, z;
    int inf;
} vo_gej;

void vo_u256_from_be(const uint8_t in[32], uint64_t out[4]);
void vo_u256_to_be(const uint64_t in[4], uint8_t out[32]);

void vo_fe_from_be(vo_fe *r, const uint8_t in[32]);
void vo_fe_to_be(const vo_fe *a, uint8_t out[32]);
int vo_fe_is_zero(const vo_fe *a);
int vo_fe_is_odd(const vo_fe *a);
int vo_fe_eq(const vo_fe *a, const vo_fe *b);
void vo_fe_add(vo_fe *r, const vo_fe *a, const vo_fe *b);
void vo_fe_sub(vo_fe *r, const vo_fe *a, const vo_fe *b);
void vo_fe_neg(vo_fe *r, const vo_fe *a);
void vo_fe_mul(vo_fe *r, const vo_fe *a, const vo_fe *b);
void vo_fe_sqr(vo_fe *r, const vo_fe *a);
void vo_fe_inv(vo_fe *r, const vo_fe *a);

void vo_ge_generator(vo_ge *g);
void vo_gej_from_ge(vo_gej *r, const vo_ge *a);
void vo_ge_from_gej(vo_ge *r, const vo_gej *a);
void vo_gej_double(vo_gej *r, const vo_gej *a);
void vo_gej_add(vo_gej *r, const vo_gej *a, const vo_gej *b);
void vo_gej_add_ge(vo_gej *r, const vo_gej *a, const vo_ge *b);
void vo_ecmult_gen(vo_gej *r, const uint64_t k[4]);
void vo_ecmult_naive(vo_gej *r, const uint64_t k[4]);
int vo_scalar_valid(const uint64_t k[4]);

#endif
