/*
 * vo_scan.c — the reference's CPU scan loops for the parity oracle / CPU baseline
 * (TEST INFRASTRUCTURE ONLY, see vgen_oracle.h).
 *
 * Restates src/scanner.rs:
 *   scan_range_cpu     :211-330  keys start..=end in 10 000-key batches spread over worker threads
 *                                (rayon there, pthreads here), FULL scalar multiplication per key
 *                                (:294), invalid keys skipped and not counted (:294-295), every
 *                                match in a batch kept — not truncated to `count` (:305-308); once
 *                                `count` matches exist no further batch is started (:250-257).
 *   scan_with_progress :81-208   independent random 32-byte keys, generate + match per key,
 *                                stop at `count` matches; operations counted per whole batch (:172).
 * The reference seeds its RNG from OS entropy (:144) and is not reproducible; the key stream here is
 * a documented, seeded build-side substitute (vgen_oracle.h).
 */
#include "vgen_oracle.h"
#include "vo_internal.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_secs(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int resolve_threads(int threads) {
    if (threads > 0) return threads;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

void vo_seed_key(uint64_t seed, uint32_t shard, uint8_t out_be[32]) {
    uint8_t buf[11 + 8 + 4 + 4];
    uint32_t redraw = 0;
    for (;;) {
        size_t n = 0;
        memcpy(buf, "vgen-mi355x", 11);
        n = 11;
        for (int i = 0; i < 8; i++) buf[n++] = (uint8_t)(seed >> (8 * i));
        for (int i = 0; i < 4; i++) buf[n++] = (uint8_t)(shard >> (8 * i));
        if (redraw) /* only ever needed if the digest reduces to 0: append a counter */
            for (int i = 0; i < 4; i++) buf[n++] = (uint8_t)(redraw >> (8 * i));
        uint8_t d[32];
        vo_sha256(buf, n, d);
        /* reduce mod n: digest < 2^256 < 2n, so one conditional subtraction suffices */
        static const uint64_t N_LIMBS[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL,
                                            0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
        uint64_t k[4];
        vo_u256_from_be(d, k);
        int ge = 1;
        for (int i = 3; i >= 0; i--) {
            if (k[i] < N_LIMBS[i]) { ge = 0; break; }
            if (k[i] > N_LIMBS[i]) break;
        }
        if (ge) {
            unsigned __int128 b = 0;
            for (int i = 0; i < 4; i++) {
                unsigned __int128 t = (unsigned __int128)k[i] - N_LIMBS[i] - (uint64_t)b;
                k[i] = (uint64_t)t;
                b = (t >> 64) & 1;
            }
        }
        if (k[0] | k[1] | k[2] | k[3]) {
            vo_u256_to_be(k, out_be);
            return;
        }
        redraw++;
    }
}

/* ---- shared state -------------------------------------------------------------------------- */

typedef struct {
    int fmt;
    const vo_regex *re;
    size_t count;
    /* range mode */
    uint8_t start[32];
    uint64_t total_keys, num_batches;
    /* random mode */
    uint64_t seed, max_keys;
    /* shared */
    pthread_mutex_t mu;
    uint64_t next_batch;
    uint64_t operations;
    vo_match *matches;
    size_t n_matches, cap_matches;
    volatile int stop;
} scan_state;

static void push_matches(scan_state *st, const vo_match *m, size_t n) {
    if (st->n_matches + n > st->cap_matches) {
        size_t nc = st->cap_matches ? st->cap_matches * 2 : 64;
        while (nc < st->n_matches + n) nc *= 2;
        st->matches = (vo_match *)realloc(st->matches, nc * sizeof(vo_match));
        st->cap_matches = nc;
    }
    memcpy(st->matches + st->n_matches, m, n * sizeof(vo_match));
    st->n_matches += n;
}

#define RANGE_BATCH 10000ULL

static void *range_worker(void *arg) {
    scan_state *st = (scan_state *)arg;
    vo_match *found = NULL;
    size_t nfound = 0, capfound = 0;
    for (;;) {
        pthread_mutex_lock(&st->mu);
        if (st->stop || st->n_matches >= st->count || st->next_batch >= st->num_batches) {
            if (st->n_matches >= st->count) st->stop = 1;
            pthread_mutex_unlock(&st->mu);
            break;
        }
        uint64_t b = st->next_batch++;
        pthread_mutex_unlock(&st->mu);

        uint64_t first = b * RANGE_BATCH;
        uint64_t n = (b == st->num_batches - 1) ? st->total_keys - first : RANGE_BATCH;
        uint64_t ops = 0;
        nfound = 0;
        for (uint64_t i = 0; i < n; i++) {
            if (st->stop) break;
            uint8_t key[32];
            if (vo_key_add_u64(st->start, first + i, key)) continue; /* past 2^256 */
            vo_generated g;
            if (!vo_generate(st->fmt, key, &g)) continue;
            ops++;
            if (vo_regex_is_match(st->re, g.address)) {
                if (nfound == capfound) {
                    capfound = capfound ? capfound * 2 : 16;
                    found = (vo_match *)realloc(found, capfound * sizeof(vo_match));
                }
                memcpy(found[nfound].key, key, 32);
                found[nfound].gen = g;
                nfound++;
            }
        }
        pthread_mutex_lock(&st->mu);
        st->operations += ops;
        if (nfound) push_matches(st, found, nfound);
        pthread_mutex_unlock(&st->mu);
    }
    free(found);
    return NULL;
}

static int cmp_match_key(const void *a, const void *b) {
    return memcmp(((const vo_match *)a)->key, ((const vo_match *)b)->key, 32);
}

static void run_workers(scan_state *st, int threads, void *(*fn)(void *)) {
    int nt = resolve_threads(threads);
    pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
    for (int i = 0; i < nt; i++) pthread_create(&tids[i], NULL, fn, st);
    for (int i = 0; i < nt; i++) pthread_join(tids[i], NULL);
    free(tids);
}

int vo_scan_range(int fmt, const char *pattern, int ci, const uint8_t start_be[32],
                  const uint8_t end_be[32], size_t count, int threads, vo_scan_result *out) {
    memset(out, 0, sizeof *out);
    vo_regex *re = vo_regex_new(pattern, ci, NULL, 0);
    if (!re) return -1;
    scan_state st;
    memset(&st, 0, sizeof st);
    st.fmt = fmt;
    st.re = re;
    st.count = count;
    memcpy(st.start, start_be, 32);
    /* total = end - start + 1, saturated to u64::MAX (scanner.rs:228-229) */
    uint64_t s[4], e[4];
    vo_u256_from_be(start_be, s);
    vo_u256_from_be(end_be, e);
    unsigned __int128 bw = 0;
    uint64_t d[4];
    for (int i = 0; i < 4; i++) {
        unsigned __int128 t = (unsigned __int128)e[i] - s[i] - (uint64_t)bw;
        d[i] = (uint64_t)t;
        bw = (t >> 64) & 1;
    }
    if (bw) {
        vo_regex_free(re);
        return -2; /* end < start */
    }
    if (d[1] | d[2] | d[3] || d[0] == UINT64_MAX)
        st.total_keys = UINT64_MAX;
    else
        st.total_keys = d[0] + 1;
    st.num_batches = st.total_keys / RANGE_BATCH + (st.total_keys % RANGE_BATCH ? 1 : 0);
    pthread_mutex_init(&st.mu, NULL);
    double t0 = now_secs();
    run_workers(&st, threads, range_worker);
    out->elapsed_secs = now_secs() - t0;
    pthread_mutex_destroy(&st.mu);
    if (st.n_matches) qsort(st.matches, st.n_matches, sizeof(vo_match), cmp_match_key);
    out->matches = st.matches;
    out->n_matches = st.n_matches;
    out->operations = st.operations;
    vo_regex_free(re);
    return 0;
}

/* ---- random mode ----------------------------------------------------------------------------- */

#define RANDOM_BATCH 10000ULL /* cpu_batch_size default, scanner.rs:107 */

typedef struct {
    scan_state *st;
    uint32_t tid;
} rnd_arg;

static void random_key(uint64_t seed, uint32_t tid, uint64_t i, uint8_t out[32]) {
    /* 24 seed bytes: the u64 little-endian, then sixteen zero bytes (the product's unseeded scans fill all 24 from the OS) */
    uint8_t buf[16 + 24 + 4 + 8];
    size_t n = 16;
    memcpy(buf, "vgen-mi355x-rand", 16);
    for (int k = 0; k < 8; k++) buf[n++] = (uint8_t)(seed >> (8 * k));
    for (int k = 0; k < 16; k++) buf[n++] = 0;
    for (int k = 0; k < 4; k++) buf[n++] = (uint8_t)(tid >> (8 * k));
    for (int k = 0; k < 8; k++) buf[n++] = (uint8_t)(i >> (8 * k));
    vo_sha256(buf, n, out);
}

static void *random_worker(void *argp) {
    rnd_arg *a = (rnd_arg *)argp;
    scan_state *st = a->st;
    uint64_t i = 0;
    for (;;) {
        pthread_mutex_lock(&st->mu);
        int done = st->stop || st->n_matches >= st->count ||
                   (st->max_keys && st->operations >= st->max_keys);
        if (!done) st->operations += RANDOM_BATCH; /* counted per whole batch, scanner.rs:172 */
        pthread_mutex_unlock(&st->mu);
        if (done) break;
        for (uint64_t j = 0; j < RANDOM_BATCH; j++, i++) {
            if (st->stop) break;
            uint8_t key[32];
            random_key(st->seed, a->tid, i, key);
            vo_generated g;
            if (!vo_generate(st->fmt, key, &g)) continue;
            if (vo_regex_is_match(st->re, g.address)) {
                vo_match m;
                memcpy(m.key, key, 32);
                m.gen = g;
                pthread_mutex_lock(&st->mu);
                if (st->n_matches < st->count) push_matches(st, &m, 1);
                if (st->n_matches >= st->count) st->stop = 1;
                pthread_mutex_unlock(&st->mu);
            }
        }
    }
    return NULL;
}

int vo_scan_random(int fmt, const char *pattern, int ci, uint64_t seed, size_t count,
                   uint64_t max_keys, int threads, vo_scan_result *out) {
    memset(out, 0, sizeof *out);
    vo_regex *re = vo_regex_new(pattern, ci, NULL, 0);
    if (!re) return -1;
    scan_state st;
    memset(&st, 0, sizeof st);
    st.fmt = fmt;
    st.re = re;
    st.count = count;
    st.seed = seed;
    st.max_keys = max_keys;
    pthread_mutex_init(&st.mu, NULL);
    int nt = resolve_threads(threads);
    pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
    rnd_arg *args = (rnd_arg *)malloc(sizeof(rnd_arg) * (size_t)nt);
    double t0 = now_secs();
    for (int i = 0; i < nt; i++) {
        args[i].st = &st;
        args[i].tid = (uint32_t)i;
        pthread_create(&tids[i], NULL, random_worker, &args[i]);
    }
    for (int i = 0; i < nt; i++) pthread_join(tids[i], NULL);
    out->elapsed_secs = now_secs() - t0;
    free(tids);
    free(args);
    pthread_mutex_destroy(&st.mu);
    out->matches = st.matches;
    out->n_matches = st.n_matches;
    out->operations = st.operations;
    vo_regex_free(re);
    return 0;
}

void vo_scan_free(vo_scan_result *r) {
    free(r->matches);
    memset(r, 0, sizeof *r);
}

/* ---- bulk payload helper for the GPU parity tests ---------------------------------------------- */

typedef struct {
    int fmt;
    const uint8_t *start;
    uint64_t lo, hi;
    int plen;
    uint8_t *out;
} seq_arg;

static void *seq_worker(void *argp) {
    seq_arg *a = (seq_arg *)argp;
    for (uint64_t i = a->lo; i < a->hi; i++) {
        uint8_t key[32], pl[32];
        uint8_t *dst = a->out + i * (uint64_t)a->plen;
        memset(dst, 0, (size_t)a->plen);
        if (vo_key_add_u64(a->start, i, key)) continue;
        if (vo_payload(a->fmt, key, pl) == a->plen) memcpy(dst, pl, (size_t)a->plen);
    }
    return NULL;
}

int vo_payload_seq(int fmt, const uint8_t start_be[32], uint64_t n, int threads, uint8_t *out) {
    int nt = resolve_threads(threads);
    if ((uint64_t)nt > n) nt = n ? (int)n : 1;
    int plen = fmt == VO_FMT_P2TR ? 32 : 20;
    pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
    seq_arg *args = (seq_arg *)malloc(sizeof(seq_arg) * (size_t)nt);
    /* make sure the fixed-base table is built before the workers race for it */
    uint8_t one[32] = {0}, tmp[65];
    one[31] = 1;
    vo_pubkey(one, tmp);
    for (int i = 0; i < nt; i++) {
        args[i].fmt = fmt;
        args[i].start = start_be;
        args[i].lo = n * (uint64_t)i / (uint64_t)nt;
        args[i].hi = n * (uint64_t)(i + 1) / (uint64_t)nt;
        args[i].plen = plen;
        args[i].out = out;
        pthread_create(&tids[i], NULL, seq_worker, &args[i]);
    }
    for (int i = 0; i < nt; i++) pthread_join(tids[i], NULL);
    free(tids);
    free(args);
    return plen;
}
