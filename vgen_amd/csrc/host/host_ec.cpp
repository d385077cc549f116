// host_ec.cpp — see host_ec.h.
#include "host_ec.h"

#include <mutex>
#include <thread>

namespace vg {

namespace {

// 64 windows x 16 digits of a 4-bit fixed-base table: tbl[w][d] = d * 16^w * G (d = 0 unused).
ge g_gen_table[64][16];
std::once_flag g_gen_once;

void build_gen_table() {
    std::vector<gej> jac;
    jac.reserve(64 * 15);
    ge g;
    ge_generator(g);
    gej base;
    gej_from_ge(base, g);
    for (int w = 0; w < 64; w++) {
        // the window base must be affine for the mixed additions below
        ge base_aff;
        ge_from_gej(base_aff, base);
        gej acc;
        gej_set_infinity(acc);
        for (int d = 1; d < 16; d++) {
            gej_add_ge(acc, acc, base_aff);
            jac.push_back(acc);
        }
        for (int k = 0; k < 4; k++) gej_double(base, base);
    }
    std::vector<ge> aff(jac.size());
    host_batch_to_affine(jac.data(), aff.data(), jac.size());
    for (int w = 0; w < 64; w++)
        for (int d = 1; d < 16; d++) g_gen_table[w][d] = aff[(size_t)w * 15 + (d - 1)];
}

}  // namespace

void host_batch_to_affine(const gej *in, ge *out, size_t n) {
    if (n == 0) return;
    std::vector<fe> pre(n);
    pre[0] = in[0].z;
    for (size_t i = 1; i < n; i++) fe_mul(pre[i], pre[i - 1], in[i].z);
    fe inv;
    fe_inv(inv, pre[n - 1]);
    for (size_t i = n; i-- > 0;) {
        fe zi;
        if (i > 0) {
            fe_mul(zi, inv, pre[i - 1]);
            fe_mul(inv, inv, in[i].z);
        } else {
            zi = inv;
        }
        fe zi2, zi3;
        fe_sqr(zi2, zi);
        fe_mul(zi3, zi2, zi);
        fe_mul(out[i].x, in[i].x, zi2);
        fe_mul(out[i].y, in[i].y, zi3);
        fe_normalize(out[i].x);
        fe_normalize(out[i].y);
    }
}

bool host_ec_mul_gen(const Scalar &k, ge &out) {
    std::call_once(g_gen_once, build_gen_table);
    gej acc;
    gej_set_infinity(acc);
    for (int w = 0; w < 64; w++) {
        uint32_t d = (k.w[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (d) gej_add_ge(acc, acc, g_gen_table[w][d]);
    }
    return ge_from_gej(out, acc);
}

namespace {

// out[i] = (first + i*step) * G for i in [0, count): one fixed-base multiplication, then a chain of mixed
// additions converted to affine 4096 at a time (one inversion per chunk).
void stride_table_range(uint64_t first, uint64_t step, uint32_t count, ge *out) {
    if (count == 0) return;
    Scalar s;
    memset(&s, 0, sizeof s);
    s.w[0] = (uint32_t)first;
    s.w[1] = (uint32_t)(first >> 32);
    ge p0, st;
    host_ec_mul_gen(s, p0);
    s.w[0] = (uint32_t)step;
    s.w[1] = (uint32_t)(step >> 32);
    host_ec_mul_gen(s, st);

    const uint32_t CHUNK = 4096;
    std::vector<gej> jac(CHUNK);
    gej cur;
    gej_from_ge(cur, p0);
    for (uint32_t done = 0; done < count;) {
        uint32_t n = count - done < CHUNK ? count - done : CHUNK;
        for (uint32_t i = 0; i < n; i++) {
            jac[i] = cur;
            gej_add_ge(cur, cur, st);
        }
        host_batch_to_affine(jac.data(), out + done, n);
        done += n;
    }
}

}  // namespace

// The offset table of a context (65 536 points at the default batch size) is most of vgen_create's time when
// built by one thread (~55 ms): the range is split over the host's cores.
void host_build_stride_table(uint64_t first, uint64_t step, uint32_t count, std::vector<ge> &out) {
    out.resize(count);
    if (count == 0) return;
    {
        Scalar one{};
        one.w[0] = 1;
        ge g;
        host_ec_mul_gen(one, g);
    }
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : nt > 16 ? 16 : nt;
    if (count < 8192) nt = 1;
    const uint32_t per = (count + nt - 1) / nt;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) {
        const uint32_t lo = t * per, hi = lo + per < count ? lo + per : count;
        if (lo >= hi) break;
        th.emplace_back([=, &out]() { stride_table_range(first + (uint64_t)lo * step, step, hi - lo, out.data() + lo); });
    }
    for (auto &x : th) x.join();
}

}  // namespace vg

namespace vg {

namespace {

// out[j] = q[j] + d for affine points, one shared inversion.  false on an exceptional pair (equal x).
bool batch_affine_add(const ge *q, const ge &d, uint32_t S, ge *out) {
    fe dx[32], pre[32];
    for (uint32_t j = 0; j < S; j++) {
        fe_sub_n(dx[j], d.x, q[j].x);
        if (fe_is_zero_any(dx[j])) return false;
        if (j == 0) pre[0] = dx[0];
        else fe_mul(pre[j], pre[j - 1], dx[j]);
    }
    fe inv;
    fe_inv(inv, pre[S - 1]);
    for (uint32_t j = S; j-- > 0;) {
        fe idx;
        if (j > 0) {
            fe_mul(idx, inv, pre[j - 1]);
            fe_mul(inv, inv, dx[j]);
        } else {
            idx = inv;
        }
        fe dy, lam, x3, t, y3;
        fe_sub_n(dy, d.y, q[j].y);
        fe_mul(lam, dy, idx);
        fe_sqr(x3, lam);
        fe_sub_n(x3, x3, q[j].x);
        fe_sub_n(x3, x3, d.x);
        fe_sub_n(t, q[j].x, x3);
        fe_mul(y3, lam, t);
        fe_sub_n(y3, y3, q[j].y);
        fe_normalize(x3);
        fe_normalize(y3);
        out[j].x = x3;
        out[j].y = y3;
    }
    return true;
}

// out[i] = a[i] + b[i] for n <= 256 pairs of affine points, one shared inversion.  false on an exceptional
// pair (equal x).
bool batch_affine_add_pairs(const ge *a, const ge *b, uint32_t n, ge *out) {
    fe dx[256], pre[256];
    if (n == 0 || n > 256) return false;
    for (uint32_t j = 0; j < n; j++) {
        fe_sub_n(dx[j], b[j].x, a[j].x);
        if (fe_is_zero_any(dx[j])) return false;
        if (j == 0) pre[0] = dx[0];
        else fe_mul(pre[j], pre[j - 1], dx[j]);
    }
    fe inv;
    fe_inv(inv, pre[n - 1]);
    for (uint32_t j = n; j-- > 0;) {
        fe idx;
        if (j > 0) {
            fe_mul(idx, inv, pre[j - 1]);
            fe_mul(inv, inv, dx[j]);
        } else {
            idx = inv;
        }
        fe dy, lam, x3, t, y3;
        fe_sub_n(dy, b[j].y, a[j].y);
        fe_mul(lam, dy, idx);
        fe_sqr(x3, lam);
        fe_sub_n(x3, x3, a[j].x);
        fe_sub_n(x3, x3, b[j].x);
        fe_sub_n(t, a[j].x, x3);
        fe_mul(y3, lam, t);
        fe_sub_n(y3, y3, a[j].y);
        fe_normalize(x3);
        fe_normalize(y3);
        out[j].x = x3;
        out[j].y = y3;
    }
    return true;
}

// Fills the look-ahead of `c` (whose q / kb are current and whose stride point is valid).
void fill_ahead(SeqBaseCache &c) {
    constexpr uint32_t K = SeqBaseCache::LOOK;
    c.ahead_n = c.ahead_pos = 0;
    if (!c.mvalid) {   // (i+1) * delta * G: a chain of mixed additions, one conversion
        gej jac[K];
        gej_from_ge(jac[0], c.dpt);
        for (uint32_t i = 1; i < K; i++) {
            gej_add_ge(jac[i], jac[i - 1], c.dpt);
            if (jac[i].inf) return;
        }
        host_batch_to_affine(jac, c.mult, K);
        c.mvalid = true;
    }
    ge a[256], b[256], o[256];
    const uint32_t S = c.S;
    for (uint32_t i = 0; i < K; i++)
        for (uint32_t j = 0; j < S; j++) {
            a[i * S + j] = c.q[j];
            b[i * S + j] = c.mult[i];
        }
    if (!batch_affine_add_pairs(a, b, K * S, o)) return;
    for (uint32_t i = 0; i < K; i++)
        for (uint32_t j = 0; j < S; j++) c.ahead[i][j] = o[i * S + j];
    c.ahead_n = K;
}

bool seq_points_full(const Scalar &kb, uint32_t S, ge *out) {
    ge base, g;
    if (!host_ec_mul_gen(kb, base)) return false;
    ge_generator(g);
    gej jac[32];
    gej_from_ge(jac[0], base);
    for (uint32_t j = 1; j < S; j++) {
        gej_add_ge(jac[j], jac[j - 1], g);
        if (jac[j].inf) return false;
    }
    host_batch_to_affine(jac, out, S);
    return true;
}

}  // namespace

bool host_seq_points(SeqBaseCache &c, const Scalar &kb, uint32_t S, ge *out) {
    bool done = false;
    bool stride_repeated = false;
    if (c.valid && c.S == S && c.ahead_pos < c.ahead_n && c.dvalid) {
        // the look-ahead holds kb_prev + delta next: hand it out if that is what is asked for
        Scalar expect;
        if (!scalar_add_u64(expect, c.kb, c.delta) && scalar_cmp(expect, kb) == 0) {
            for (uint32_t j = 0; j < S; j++) out[j] = c.q[j] = c.ahead[c.ahead_pos][j];
            c.ahead_pos++;
            c.kb = kb;
            return true;
        }
        c.ahead_n = c.ahead_pos = 0;   // the walk changed course
    }
    if (c.valid && c.S == S && scalar_cmp(kb, c.kb) > 0) {
        // diff = kb - cache.kb, must fit 64 bits
        Scalar diff;
        int64_t b = 0;
        for (int i = 0; i < 8; i++) {
            int64_t t = (int64_t)kb.w[i] - c.kb.w[i] + b;
            diff.w[i] = (uint32_t)t;
            b = t >> 32;
        }
        bool small = true;
        for (int i = 2; i < 8; i++) small = small && diff.w[i] == 0;
        if (small) {
            uint64_t d = ((uint64_t)diff.w[1] << 32) | diff.w[0];
            if (!c.dvalid || c.delta != d) {
                Scalar ds{};
                ds.w[0] = diff.w[0];
                ds.w[1] = diff.w[1];
                c.dvalid = host_ec_mul_gen(ds, c.dpt);
                c.delta = d;
                c.mvalid = false;
            } else {
                stride_repeated = true;
            }
            if (c.dvalid) done = batch_affine_add(c.q, c.dpt, S, out);
        }
    }
    if (!done && !seq_points_full(kb, S, out)) {
        c.valid = false;
        return false;
    }
    c.valid = true;
    c.S = S;
    c.kb = kb;
    for (uint32_t j = 0; j < S; j++) c.q[j] = out[j];
    c.ahead_n = c.ahead_pos = 0;
    if (done && stride_repeated && SeqBaseCache::LOOK * S <= 256) fill_ahead(c);
    return true;
}

}  // namespace vg

namespace vg {
void host_gen_table_limbs(std::vector<uint32_t> &out) {
    std::call_once(g_gen_once, build_gen_table);
    out.resize((size_t)64 * 15 * 18);
    for (int w = 0; w < 64; w++)
        for (int d = 1; d < 16; d++) {
            uint32_t *p = &out[((size_t)w * 15 + (d - 1)) * 18];
            for (int i = 0; i < 9; i++) {
                p[i] = g_gen_table[w][d].x.n[i];
                p[9 + i] = g_gen_table[w][d].y.n[i];
            }
        }
}
// 8-bit windows: entry (w, d) = d * 256^w * G = lo(d) * 16^(2w) * G + hi(d) * 16^(2w+1) * G, the sum of two
// entries of the 4-bit table; 8160 additions, one shared inversion.
// (Windows are independent: a few threads take four windows each, with an inversion per thread — 5.4 ms on one thread,
// part of what the first P2TR / arbitrary-scalar dispatch of a context waits for.)
void host_gen_table8_limbs(std::vector<uint32_t> &out) {
    std::call_once(g_gen_once, build_gen_table);
    out.assign((size_t)32 * 255 * 20, 0u);
    auto windows = [&out](int w0, int w1) {
        const size_t n = (size_t)(w1 - w0) * 255;
        std::vector<gej> jac(n);
        for (int w = w0; w < w1; w++)
            for (int d = 1; d < 256; d++) {
                const int lo = d & 15, hi = d >> 4;
                gej acc;
                if (lo) {
                    gej_from_ge(acc, g_gen_table[2 * w][lo]);
                    if (hi) gej_add_ge(acc, acc, g_gen_table[2 * w + 1][hi]);
                } else {
                    gej_from_ge(acc, g_gen_table[2 * w + 1][hi]);
                }
                jac[(size_t)(w - w0) * 255 + (d - 1)] = acc;
            }
        std::vector<ge> aff(n);
        host_batch_to_affine(jac.data(), aff.data(), n);
        for (size_t e = 0; e < n; e++) {
            const size_t o = ((size_t)w0 * 255 + e) * 20;
            for (int i = 0; i < 9; i++) {
                out[o + i] = aff[e].x.n[i];
                out[o + 9 + i] = aff[e].y.n[i];
            }
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt >= 8 ? 8 : nt >= 4 ? 4 : nt >= 2 ? 2 : 1;
    const int per = 32 / (int)nt;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(windows, (int)t * per, (int)(t + 1) * per);
    windows(0, per);
    for (auto &x : th) x.join();
}
}  // namespace vg
