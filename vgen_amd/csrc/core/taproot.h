// taproot.h — BIP-341 key-path output key on top of ec.h / hash.h; single source (device + host).
//
// What the reference does on the HOST for every key of a P2TR batch (XOnlyPublicKey::from_slice +
// Address::p2tr(secp, internal_key, None, ..), src/gpu.rs:1287-1291, "CPU bound by design"
// src/shaders/search_p2tr.wgsl:112): P = lift_x(x) (even Y), t = TapTweak(x), Q = P + t*G, output x(Q).
// Here it runs per lane on the device: the tweak multiplication walks the 8-bit fixed-window generator table
// (global memory, L2-resident), and the final 1/Z is shared by the whole workgroup (kernels.hip).
#pragma once
#include "ec.h"
#include "hash.h"

namespace vg {

constexpr u32 TAP_ORDER_N[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

// t = TapTweak(x) as eight little-endian words; xw: x as eight words, xw[0] least significant.  Returns false (and t = 1, a
// harmless stand-in) when the tweak is not a valid scalar (t == 0 or t >= n: probability ~2^-128) — such keys yield no
// address (Address::p2tr would fail there).
VG_HD bool taproot_tweak_scalar(const u32 xw[8], u32 k[8]) {
    u32 tb[8];
    sha256_taptweak(xw, tb);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = tb[7 - i];   // little-endian words
    u32 nz = 0;
    int cmp = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        nz |= k[i];
        const int d = (k[i] > TAP_ORDER_N[i]) - (k[i] < TAP_ORDER_N[i]);
        cmp = cmp == 0 ? d : cmp;
    }
    const bool ok = nz != 0 && cmp < 0;
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = 0;
        k[0] = 1;
    }
    return ok;
}

// lift_x: the point with this x and EVEN y (y canonical).
VG_HD void taproot_lift_even(ge &p, const fe &x, const fe &y) {
    p.x = x;
    fe ny;
    fe_neg(ny, y, 1);
    fe_normalize(ny);
    const bool odd = (y.n[0] & 1u) != 0;
#pragma unroll
    for (int i = 0; i < 9; i++) p.y.n[i] = odd ? ny.n[i] : y.n[i];
}

// x, y: canonical affine internal key; tabs: the generator tables (ec.h).  q = lift_x(x) + TapTweak(x)*G in Jacobian
// coordinates.  Returns false when the tweak is not a valid scalar (q is then some other valid point).
// q.z == 0 (t*G == -P) is the caller's to check.
VG_HD bool taproot_tweak_point(const fe &x, const fe &y, const GenTables &tabs, gej &q) {
    u32 xw[8], k[8];
    fe_to_words(x, xw);
    const bool ok = taproot_tweak_scalar(xw, k);
    gej tg;
    ec_mul_gen_tables(tg, k, tabs);
    ge p;
    taproot_lift_even(p, x, y);
    gej_add_ge_nz(q, tg, p);              // t*G == +/-P would need t = +/-d: negligible; Z = 0 then
    return ok;
}

// True when z represents 0 mod p (z weakly normalised).
VG_HD bool taproot_z_is_zero(const fe &z) {
    fe zc = z;
    fe_canonicalize(zc);
    return fe_is_zero_canonical(zc);
}

// x(Q) = X / Z^2 as eight words (out_xw[0] least significant), given zi = 1/Z.
VG_HD void taproot_affine_x(const gej &q, const fe &zi, u32 out_xw[8]) {
    fe zi2, qx;
    fe_sqr(zi2, zi);
    fe_mul(qx, q.x, zi2);
    fe_canonicalize(qx);
    fe_to_words(qx, out_xw);
}

// Single-key form (host: vgen_derive, match confirmation, tests): its own inversion.
VG_HD bool taproot_output_x(const fe &x, const fe &y, const u32 *tab8, u32 out_xw[8]) {
    const GenTables tabs{tab8, nullptr, 0};
    gej q;
    const bool ok = taproot_tweak_point(x, y, tabs, q);
    const bool inf = taproot_z_is_zero(q.z);
    fe zi;
    fe_inv(zi, q.z);
    taproot_affine_x(q, zi, out_xw);
    return ok && !inf;
}

}  // namespace vg
