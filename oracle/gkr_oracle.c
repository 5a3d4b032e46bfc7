/* gkr_oracle.c -- CPU restatement of the Consensys/gkr-mimc hot path.  TEST INFRASTRUCTURE ONLY
 * (see gkr_oracle.h for the rules and the pinning status).  Plain C + OpenMP.
 *
 * Reference citations are file:line relative to the reference checkout. */
#include "gkr_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ---- field constants (q also in hash/ark.go:7) ---------------------------- */
static const u64 Qm[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 QINV = 0xc2e1f593efffffffULL;                 /* -q^-1 mod 2^64 */
static const ofr_t ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
static const ofr_t R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};

static const ofr_t ARKS[100] = {
#include "arks.inc"
};

#define MIMC_ROUNDS 91 /* hash/mimc.go:8 */

static int g_threads = 0;
int oracle_num_threads(void) {
    if (g_threads > 0) return g_threads;
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

/* ---- gnark-crypto fr arithmetic (external; CIOS Montgomery, canonical outputs) ---------------- */
static inline int geq_q(const u64 t[4]) {
    for (int i = 3; i >= 0; i--) {
        if (t[i] > Qm[i]) return 1;
        if (t[i] < Qm[i]) return 0;
    }
    return 1;
}
static inline void sub_q(u64 t[4]) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)t[i] - Qm[i] - (u64)b;
        t[i] = (u64)d;
        b = (d >> 64) & 1;
    }
}
static inline void fr_mul_generic(ofr_t *z, const ofr_t *x, const ofr_t *y) {
    u64 t[4] = {0, 0, 0, 0};
    u64 t4 = 0;
    for (int i = 0; i < 4; i++) {
        u128 acc;
        u64 c = 0;
        for (int j = 0; j < 4; j++) {
            acc = (u128)x->l[j] * y->l[i] + t[j] + c;
            t[j] = (u64)acc;
            c = (u64)(acc >> 64);
        }
        acc = (u128)t4 + c;
        t4 = (u64)acc;
        u64 t5 = (u64)(acc >> 64);
        u64 m = t[0] * QINV;
        acc = (u128)m * Qm[0] + t[0];
        c = (u64)(acc >> 64);
        for (int j = 1; j < 4; j++) {
            acc = (u128)m * Qm[j] + t[j] + c;
            t[j - 1] = (u64)acc;
            c = (u64)(acc >> 64);
        }
        acc = (u128)t4 + c;
        t[3] = (u64)acc;
        t4 = t5 + (u64)(acc >> 64);
    }
    if (t4 || geq_q(t)) sub_q(t);
    memcpy(z->l, t, 32);
}
/* fr.Element.Mul.  gnark-crypto's amd64 Mul is the "no-carry" CIOS (the top word of q is below 2^63, so the running value
 * never needs a fifth word) written with MULX/ADCX/ADOX.  This is the same algorithm with the four rows unrolled (built
 * with -mbmi2 -madx gcc emits mulx and keeps the rows in registers); -DORACLE_GENERIC_MUL selects the looped five-word CIOS
 * of rounds 1-2, kept as fr_mul_generic and compared in the tests.  Measured in isolation (oracle_bench_fr_mul, one core)
 * both take the same time -- gcc unrolls the loops of the generic form as well -- so the multiplication is NOT what
 * separates this port from the Go binary; bench.py reports the isolated figure beside the whole-prover one.  (A
 * branch-free conditional subtraction was tried for Mul/Add/Sub and is slower with gcc: 28 against 18.5 ns per product.) */
#define OFR_ROW(yi)                                                            \
    {                                                                          \
        u128 a = (u128)x->l[0] * (yi) + t0;                                    \
        u64 lo0 = (u64)a;                                                      \
        a = (u128)x->l[1] * (yi) + t1 + (u64)(a >> 64);                        \
        u64 lo1 = (u64)a;                                                      \
        a = (u128)x->l[2] * (yi) + t2 + (u64)(a >> 64);                        \
        u64 lo2 = (u64)a;                                                      \
        a = (u128)x->l[3] * (yi) + t3 + (u64)(a >> 64);                        \
        u64 lo3 = (u64)a;                                                      \
        u64 hi = (u64)(a >> 64);                                               \
        u64 m = lo0 * QINV;                                                    \
        a = (u128)m * Qm[0] + lo0;                                             \
        a = (u128)m * Qm[1] + lo1 + (u64)(a >> 64);                            \
        t0 = (u64)a;                                                           \
        a = (u128)m * Qm[2] + lo2 + (u64)(a >> 64);                            \
        t1 = (u64)a;                                                           \
        a = (u128)m * Qm[3] + lo3 + (u64)(a >> 64);                            \
        t2 = (u64)a;                                                           \
        t3 = hi + (u64)(a >> 64); /* q < 2^254: no overflow (no-carry CIOS) */ \
    }
static inline void fr_mul_nocarry(ofr_t *z, const ofr_t *x, const ofr_t *y) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    const u64 y0 = y->l[0], y1 = y->l[1], y2 = y->l[2], y3 = y->l[3];
    OFR_ROW(y0)
    OFR_ROW(y1)
    OFR_ROW(y2)
    OFR_ROW(y3)
    u64 t[4] = {t0, t1, t2, t3};
    if (geq_q(t)) sub_q(t);
    memcpy(z->l, t, 32);
}
#undef OFR_ROW
#ifdef ORACLE_GENERIC_MUL
#define fr_mul fr_mul_generic
#else
#define fr_mul fr_mul_nocarry
#endif
#if defined(__x86_64__)
/* add-with-carry chains and a branch-free select (adc/sbb + cmov): the outcome of the conditional subtraction is a coin
 * flip for sums of random elements, and a mispredicted branch costs more than the four subtractions */
static inline void fr_add(ofr_t *z, const ofr_t *x, const ofr_t *y) {
    unsigned long long t0, t1, t2, t3, d0, d1, d2, d3;
    unsigned char c = _addcarry_u64(0, x->l[0], y->l[0], &t0);
    c = _addcarry_u64(c, x->l[1], y->l[1], &t1);
    c = _addcarry_u64(c, x->l[2], y->l[2], &t2);
    (void)_addcarry_u64(c, x->l[3], y->l[3], &t3); /* q < 2^254: no carry out of limb 3 */
    unsigned char b = _subborrow_u64(0, t0, Qm[0], &d0);
    b = _subborrow_u64(b, t1, Qm[1], &d1);
    b = _subborrow_u64(b, t2, Qm[2], &d2);
    b = _subborrow_u64(b, t3, Qm[3], &d3);
    z->l[0] = b ? t0 : d0;
    z->l[1] = b ? t1 : d1;
    z->l[2] = b ? t2 : d2;
    z->l[3] = b ? t3 : d3;
}
static inline void fr_sub(ofr_t *z, const ofr_t *x, const ofr_t *y) {
    unsigned long long t0, t1, t2, t3;
    unsigned char b = _subborrow_u64(0, x->l[0], y->l[0], &t0);
    b = _subborrow_u64(b, x->l[1], y->l[1], &t1);
    b = _subborrow_u64(b, x->l[2], y->l[2], &t2);
    b = _subborrow_u64(b, x->l[3], y->l[3], &t3);
    const u64 m = (u64)0 - (u64)b;            /* + q when the subtraction borrowed */
    unsigned char c = _addcarry_u64(0, t0, Qm[0] & m, &t0);
    c = _addcarry_u64(c, t1, Qm[1] & m, &t1);
    c = _addcarry_u64(c, t2, Qm[2] & m, &t2);
    (void)_addcarry_u64(c, t3, Qm[3] & m, &t3);
    z->l[0] = t0;
    z->l[1] = t1;
    z->l[2] = t2;
    z->l[3] = t3;
}
#else
static inline void fr_add(ofr_t *z, const ofr_t *x, const ofr_t *y) {
    u64 t[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)x->l[i] + y->l[i];
        t[i] = (u64)c;
        c >>= 64;
    }
    if (geq_q(t)) sub_q(t); /* q < 2^254: no carry out of limb 3 */
    memcpy(z->l, t, 32);
}
static inline void fr_sub(ofr_t *z, const ofr_t *x, const ofr_t *y) {
    u64 t[4];
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x->l[i] - y->l[i] - (u64)b;
        t[i] = (u64)d;
        b = (d >> 64) & 1;
    }
    if (b) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + Qm[i];
            t[i] = (u64)c;
            c >>= 64;
        }
    }
    memcpy(z->l, t, 32);
}
#endif
static inline int fr_eq(const ofr_t *a, const ofr_t *b) { return memcmp(a, b, 32) == 0; }
static inline int fr_is_zero(const ofr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

void oracle_fr_mul(ofr_t *o, const ofr_t *a, const ofr_t *b) { fr_mul(o, a, b); }
void oracle_fr_mul_generic(ofr_t *o, const ofr_t *a, const ofr_t *b) { fr_mul_generic(o, a, b); }
/* ns per multiplication on one core: a chain of n dependent products (latency) and n independent ones (throughput) */
void oracle_bench_fr_mul(long n, int generic, double *ns_dependent, double *ns_independent) {
    struct timespec t0, t1;
    volatile ofr_t va = R2, vb = ARKS[3];
    ofr_t a = va, b = vb, acc[8];
    for (int k = 0; k < 8; k++) acc[k] = ARKS[k];
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (generic) for (long i = 0; i < n; i++) fr_mul_generic(&a, &a, &b);
    else for (long i = 0; i < n; i++) fr_mul_nocarry(&a, &a, &b);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *ns_dependent = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / (double)n;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long i = 0; i < n / 8; i++)
        for (int k = 0; k < 8; k++) {
            if (generic) fr_mul_generic(&acc[k], &acc[k], &a);
            else fr_mul_nocarry(&acc[k], &acc[k], &a);
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *ns_independent = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / (double)(n / 8 * 8);
    volatile u64 sink = a.l[0];
    for (int k = 0; k < 8; k++) sink ^= acc[k].l[0];
    (void)sink;
}
void oracle_fr_add(ofr_t *o, const ofr_t *a, const ofr_t *b) { fr_add(o, a, b); }
void oracle_fr_sub(ofr_t *o, const ofr_t *a, const ofr_t *b) { fr_sub(o, a, b); }
void oracle_fr_from_u64(ofr_t *out, u64 v) { /* SetUint64: v * R^2 * R^-1 */
    ofr_t t = {{v, 0, 0, 0}};
    fr_mul(out, &t, &R2);
}
void oracle_fr_from_regular(ofr_t *out, const u64 in[4]) {
    ofr_t t;
    memcpy(t.l, in, 32);
    fr_mul(out, &t, &R2);
}
void oracle_fr_to_regular(u64 out[4], const ofr_t *a) {
    ofr_t one = {{1, 0, 0, 0}}, t;
    fr_mul(&t, a, &one);
    memcpy(out, t.l, 32);
}
void oracle_fr_inverse(ofr_t *out, const ofr_t *a) { /* a^(q-2); Inverse(0) = 0 as gnark-crypto */
    u64 e[4];
    memcpy(e, Qm, 32);
    e[0] -= 2;
    ofr_t res = ONE, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&res, &res, &base);
        fr_mul(&base, &base, &base);
    }
    *out = res;
}

/* ---- hash/mimc.go, hash/poseidon.go:129-135, common/ ------------------------------------------ */
static inline void sbox(ofr_t *x) { /* x^7: sq, mul, sq, mul */
    ofr_t t = *x;
    fr_mul(x, x, x);
    fr_mul(x, x, &t);
    fr_mul(x, x, x);
    fr_mul(x, x, &t);
}
void oracle_mimc_keyed_permutation(ofr_t *out, const ofr_t *x, const ofr_t *key) { /* hash/mimc.go:31-39 */
    ofr_t res = *x;
    for (int i = 0; i < MIMC_ROUNDS; i++) {
        fr_add(&res, &res, key);
        fr_add(&res, &res, &ARKS[i]);
        sbox(&res);
    }
    *out = res;
}
void oracle_mimc_hash(ofr_t *out, const ofr_t *in, size_t n) { /* hash/mimc.go:11-28,43-49 */
    ofr_t state = {{0, 0, 0, 0}};
    for (size_t k = 0; k < n; k++) {
        ofr_t ns;
        oracle_mimc_keyed_permutation(&ns, &in[k], &state);
        fr_add(&ns, &ns, &state);     /* MimcBlockCipher: + key */
        fr_add(&state, &state, &ns);  /* state += newState */
        fr_add(&state, &state, &in[k]);
    }
    *out = state;
}
void oracle_random_fr_array(ofr_t *out, size_t n) { /* common/common.go:49-55 */
    for (size_t i = 0; i < n; i++) oracle_fr_from_u64(&out[i], ((u64)i * (u64)i) ^ 0xf45c9df123fULL);
}
void oracle_get_ark(ofr_t *out, int i) { *out = ARKS[i]; }

/* ---- poly/multilin.go ------------------------------------------------------------------------- */
static void fold_chunk(ofr_t *tbl, size_t len, const ofr_t *r, size_t start, size_t stop) { /* :26-36 */
    size_t mid = len / 2;
    ofr_t *bottom = tbl, *top = tbl + mid;
    for (size_t i = start; i < stop; i++) {
        fr_sub(&top[i], &top[i], &bottom[i]);
        fr_mul(&top[i], &top[i], r);
        fr_add(&bottom[i], &bottom[i], &top[i]);
    }
}
void oracle_fold(ofr_t *tbl, size_t len, const ofr_t *r) { /* :19-23 */
    size_t mid = len / 2;
#pragma omp parallel for schedule(static) if (mid >= 2048)
    for (size_t c = 0; c < (mid + 1023) / 1024; c++) {
        size_t s = c * 1024, e = s + 1024 < mid ? s + 1024 : mid;
        fold_chunk(tbl, len, r, s, e);
    }
}
void oracle_evaluate(ofr_t *out, const ofr_t *tbl, size_t len, const ofr_t *coords, int n) { /* :59-66 */
    ofr_t *cp = (ofr_t *)malloc(len * sizeof(ofr_t));
    memcpy(cp, tbl, len * sizeof(ofr_t));
    size_t cur = len;
    for (int i = 0; i < n; i++) {
        oracle_fold(cp, cur, &coords[i]);
        cur /= 2;
    }
    *out = cp[0];
    free(cp);
}

/* ---- poly/eq.go -------------------------------------------------------------------------------- */
void oracle_eval_eq(ofr_t *out, const ofr_t *q, const ofr_t *h, int n) { /* :19-32 */
    ofr_t res = ONE, nxt, sum;
    for (int i = 0; i < n; i++) {
        fr_mul(&nxt, &q[i], &h[i]);
        fr_add(&nxt, &nxt, &nxt);
        fr_add(&nxt, &nxt, &ONE);
        fr_add(&sum, &q[i], &h[i]);
        fr_sub(&nxt, &nxt, &sum);
        fr_mul(&res, &res, &nxt);
    }
    *out = res;
}
void oracle_folded_eq_table(ofr_t *t, const ofr_t *q, int n, const ofr_t *mult) { /* :41-59 */
    t[0] = mult ? *mult : ONE;
    for (int i = 0; i < n; i++) {
        for (size_t j = 0; j < ((size_t)1 << i); j++) {
            size_t J = j << (n - i);
            size_t JN = J + ((size_t)1 << (n - 1 - i));
            fr_mul(&t[JN], &q[i], &t[J]);
            fr_sub(&t[J], &t[J], &t[JN]);
        }
    }
}
static int log2_ceil(size_t a) { /* common/math.go:19-36 */
    int f = 0;
    for (size_t i = a; i > 1; i >>= 1) f++;
    if (a != ((size_t)1 << f)) f++;
    return f;
}
void oracle_chunk_of_eq_table(ofr_t *out, size_t chunk_id, size_t chunk_size, const ofr_t *q, int n,
                              const ofr_t *mult) { /* :62-89 */
    size_t n_chunks = ((size_t)1 << n) / chunk_size;
    int lg = log2_ceil(n_chunks);
    ofr_t r = mult ? *mult : ONE, tmp;
    for (int k = 0; k < lg; k++) {
        const ofr_t *rho = &q[lg - k - 1];
        if ((chunk_id >> k) & 1) {
            fr_mul(&r, &r, rho);
        } else {
            fr_sub(&tmp, &ONE, rho);
            fr_mul(&r, &r, &tmp);
        }
    }
    oracle_folded_eq_table(out + chunk_id * chunk_size, q + lg, n - lg, &r);
}

/* ---- poly/lagrange.go -------------------------------------------------------------------------- */
void oracle_eval_univariate(ofr_t *out, const ofr_t *c, int n, const ofr_t *x) { /* :31-39 */
    ofr_t res = c[n - 1];
    for (int i = n - 2; i >= 0; i--) {
        fr_mul(&res, &res, x);
        fr_add(&res, &res, &c[i]);
    }
    *out = res;
}
#define MAX_DOMAIN 12 /* :21 */
void oracle_lagrange_coefficient(ofr_t *out, int domain) { /* :42-92 */
    ofr_t zero = {{0, 0, 0, 0}};
    ofr_t bin0[MAX_DOMAIN + 1];
    for (int i = 0; i < domain; i++) {
        ofr_t v;
        oracle_fr_from_u64(&v, (u64)i);
        fr_sub(&bin0[i], &zero, &v);
    }
    for (int l = 0; l < domain; l++) {
        ofr_t acc[MAX_DOMAIN + 1], upd[MAX_DOMAIN + 1], tmp;
        for (int j = 0; j < domain; j++) acc[j] = zero;
        acc[0] = ONE;
        for (int i = 0; i < domain; i++) {
            if (i == l) continue;
            for (int j = 0; j < domain; j++) upd[j] = zero;
            for (int j = 0; j < domain; j++) {
                int kmax = 2 < domain - j ? 2 : domain - j;
                for (int k = 0; k < kmax; k++) {
                    fr_mul(&tmp, &acc[j], k == 0 ? &bin0[i] : &ONE);
                    fr_add(&upd[j + k], &upd[j + k], &tmp);
                }
            }
            memcpy(acc, upd, sizeof(ofr_t) * domain);
        }
        ofr_t lf, norm;
        oracle_fr_from_u64(&lf, (u64)l);
        oracle_eval_univariate(&norm, acc, domain, &lf);
        oracle_fr_inverse(&norm, &norm);
        for (int j = 0; j < domain; j++) fr_mul(&out[l * domain + j], &acc[j], &norm);
    }
}
static ofr_t g_lagrange[MAX_DOMAIN + 1][MAX_DOMAIN * MAX_DOMAIN];
static int g_lagrange_ready = 0;
static void init_lagrange(void) { /* :10-29 */
#pragma omp critical(oracle_lagrange)
    {
        if (!g_lagrange_ready) {
            for (int d = 1; d <= MAX_DOMAIN; d++) oracle_lagrange_coefficient(g_lagrange[d], d);
            g_lagrange_ready = 1;
        }
    }
}
int oracle_interpolate_on_range(ofr_t *out, const ofr_t *values, int n) { /* :96-111 */
    if (n < 1 || n > MAX_DOMAIN) return -1;
    if (!g_lagrange_ready) init_lagrange();
    ofr_t tmp;
    for (int j = 0; j < n; j++) memset(&out[j], 0, sizeof(ofr_t));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            fr_mul(&tmp, &g_lagrange[n][i * n + j], &values[i]);
            fr_add(&out[j], &out[j], &tmp);
        }
    return 0;
}

/* ---- circuit/gates ------------------------------------------------------------------------------ */
static inline void gate_eval(int gate, const ofr_t *ark, ofr_t *res, const ofr_t *const *xs) {
    if (gate == ORACLE_GATE_CIPHER) { /* circuit/gates/cipher.go:45-55 */
        ofr_t tmp;
        fr_add(&tmp, xs[1], ark);
        fr_add(&tmp, &tmp, xs[0]);
        fr_mul(res, &tmp, &tmp);
        fr_mul(res, res, &tmp);
        fr_mul(res, res, res);
        fr_mul(res, res, &tmp);
    } else if (gate == ORACLE_GATE_ADD) { /* build-defined: xs[0] + xs[1] + Ark (GMiMC's linear branches, hash/gmimc.go:52-58) */
        fr_add(res, xs[0], xs[1]);
        fr_add(res, res, ark);
    } else { /* circuit/gates/copy.go:20-22 */
        *res = *xs[0];
    }
}
/* the variadic build-defined gates: the arity is the layer's len(In) */
static inline void gate_eval_n(int gate, int arity, const ofr_t *ark, ofr_t *res, const ofr_t *const *xs) {
    if (gate != ORACLE_GATE_SUM && gate != ORACLE_GATE_SUM_POW7) {
        gate_eval(gate, ark, res, xs);
        return;
    }
    ofr_t s = *ark;
    for (int k = 0; k < arity; k++) fr_add(&s, &s, xs[k]);
    if (gate == ORACLE_GATE_SUM) {
        *res = s;
    } else { /* same square-and-multiply chain as cipher.go:36-40 */
        fr_mul(res, &s, &s);
        fr_mul(res, res, &s);
        fr_mul(res, res, res);
        fr_mul(res, res, &s);
    }
}
static int gate_degree(int gate) { /* cipher.go:68-70, copy.go:30-32; add / sum: linear */
    return gate == ORACLE_GATE_CIPHER || gate == ORACLE_GATE_SUM_POW7 ? 7 : 1;
}

void oracle_gate_eval_batch(int gate, const ofr_t *ark, ofr_t *res, const ofr_t *const *xs, int arity, size_t n) {
    /* cipher.go:25-42 / copy.go:15-17 */
    if (gate == ORACLE_GATE_CIPHER) {
        for (size_t i = 0; i < n; i++) {
            ofr_t tmp;
            fr_add(&tmp, &xs[1][i], ark);
            fr_add(&tmp, &tmp, &xs[0][i]);
            fr_mul(&res[i], &tmp, &tmp);
            fr_mul(&res[i], &res[i], &tmp);
            fr_mul(&res[i], &res[i], &res[i]);
            fr_mul(&res[i], &res[i], &tmp);
        }
    } else if (gate == ORACLE_GATE_ADD) {
        for (size_t i = 0; i < n; i++) {
            fr_add(&res[i], &xs[0][i], &xs[1][i]);
            fr_add(&res[i], &res[i], ark);
        }
    } else if (gate == ORACLE_GATE_SUM || gate == ORACLE_GATE_SUM_POW7) {
        for (size_t i = 0; i < n; i++) {
            const ofr_t *at[ORACLE_MAX_GATE_INPUTS];
            for (int k = 0; k < arity; k++) at[k] = &xs[k][i];
            gate_eval_n(gate, arity, ark, &res[i], at);
        }
    } else {
        memcpy(res, xs[0], n * sizeof(ofr_t));
    }
}

/* ---- sumcheck ------------------------------------------------------------------------------------ */
#define EVAL_SUBCHUNK 128  /* sumcheck/algo.go:9 */
#define MAX_ARITY 4
#define MAX_EVALS 12

typedef struct {
    ofr_t *X[MAX_ARITY];
    ofr_t *Eq;
    size_t len; /* current table length */
    int arity, gate, degree;
    ofr_t ark;
} instance_t; /* sumcheck/instance.go:12-18 */

/* sumcheck/algo.go:54-205 */
static void get_partial_poly_chunk(const instance_t *inst, size_t start, size_t stop, ofr_t *evals) {
    int n_evals = inst->degree + 1, n_in = inst->arity;
    size_t mid = inst->len / 2;
    ofr_t tmpEvals[EVAL_SUBCHUNK], tmpEqs[EVAL_SUBCHUNK], dEqs[EVAL_SUBCHUNK];
    ofr_t tmpXs[EVAL_SUBCHUNK * MAX_ARITY], dXs[EVAL_SUBCHUNK * MAX_ARITY];
    const ofr_t *buf[MAX_ARITY];
    ofr_t v;
    for (int t = 0; t < n_evals; t++) memset(&evals[t], 0, sizeof(ofr_t));

    for (size_t s0 = start; s0 < stop; s0 += EVAL_SUBCHUNK) {
        size_t s1 = s0 + EVAL_SUBCHUNK < stop ? s0 + EVAL_SUBCHUNK : stop;
        size_t len = s1 - s0;
        /* t = 0 (:101-126) */
        for (int k = 0; k < n_in; k++) buf[k] = inst->X[k] + s0;
        oracle_gate_eval_batch(inst->gate, &inst->ark, tmpEvals, buf, n_in, len);
        for (size_t x = 0; x < len; x++) {
            fr_mul(&v, &inst->Eq[s0 + x], &tmpEvals[x]);
            fr_add(&evals[0], &evals[0], &v);
        }
        /* t = 1 (:128-149) */
        for (int k = 0; k < n_in; k++) buf[k] = inst->X[k] + s0 + mid;
        oracle_gate_eval_batch(inst->gate, &inst->ark, tmpEvals, buf, n_in, len);
        for (size_t x = 0; x < len; x++) {
            fr_mul(&v, &inst->Eq[s0 + mid + x], &tmpEvals[x]);
            fr_add(&evals[1], &evals[1], &v);
        }
        /* t >= 2 (:151-201) */
        memcpy(tmpEqs, inst->Eq + s0 + mid, len * sizeof(ofr_t));
        for (size_t x = 0; x < len; x++) fr_sub(&dEqs[x], &inst->Eq[s0 + mid + x], &inst->Eq[s0 + x]);
        for (int k = 0; k < n_in; k++) {
            size_t off = (size_t)k * len;
            for (size_t x = 0; x < len; x++) fr_sub(&dXs[off + x], &inst->X[k][s0 + mid + x], &inst->X[k][s0 + x]);
            memcpy(tmpXs + off, inst->X[k] + s0 + mid, len * sizeof(ofr_t));
            buf[k] = tmpXs + off;
        }
        for (int t = 2; t < n_evals; t++) {
            for (size_t x = 0; x < len; x++) fr_add(&tmpEqs[x], &tmpEqs[x], &dEqs[x]);
            for (size_t kx = 0; kx < (size_t)n_in * len; kx++) fr_add(&tmpXs[kx], &tmpXs[kx], &dXs[kx]);
            oracle_gate_eval_batch(inst->gate, &inst->ark, tmpEvals, buf, n_in, len);
            for (size_t x = 0; x < len; x++) {
                fr_mul(&v, &tmpEqs[x], &tmpEvals[x]);
                fr_add(&evals[t], &evals[t], &v);
            }
        }
    }
}

/* common/parallelize.go:49-88 (TryDispatch) restated as a task splitter: returns #tasks (0 = run inline) */
static size_t try_dispatch(size_t n_iter, size_t min_task, size_t *per_task, size_t *extra) {
    size_t nb_tasks = (size_t)oracle_num_threads() * 8;
    size_t per = n_iter / nb_tasks;
    if (per < min_task) {
        per = min_task;
        nb_tasks = n_iter / per;
    }
    if (nb_tasks <= 1) return 0;
    *per_task = per;
    *extra = n_iter - nb_tasks * per;
    return nb_tasks;
}
static inline void task_range(size_t i, size_t per, size_t extra, size_t *start, size_t *stop) {
    size_t off = i < extra ? i : extra;
    *start = i * per + off;
    *stop = *start + per + (i < extra ? 1 : 0);
}

/* sumcheck/prover.go:148-163 + 236-245 */
static void dispatch_partial_evals(const instance_t *inst, ofr_t *evals) {
    size_t mid = inst->len / 2, per = 0, extra = 0;
    int n_evals = inst->degree + 1;
    size_t nt = try_dispatch(mid, 64, &per, &extra);
    if (nt < 1) {
        get_partial_poly_chunk(inst, 0, mid, evals);
        return;
    }
    for (int t = 0; t < n_evals; t++) memset(&evals[t], 0, sizeof(ofr_t));
#pragma omp parallel
    {
        ofr_t local[MAX_EVALS], acc[MAX_EVALS];
        memset(acc, 0, sizeof(acc));
#pragma omp for schedule(dynamic, 1) nowait
        for (size_t i = 0; i < nt; i++) {
            size_t s, e;
            task_range(i, per, extra, &s, &e);
            get_partial_poly_chunk(inst, s, e, local);
            for (int t = 0; t < n_evals; t++) fr_add(&acc[t], &acc[t], &local[t]);
        }
#pragma omp critical(oracle_evals)
        for (int t = 0; t < n_evals; t++) fr_add(&evals[t], &evals[t], &acc[t]);
    }
}

/* sumcheck/prover.go:167-190 + algo.go:46-51 */
static void dispatch_folding(instance_t *inst, const ofr_t *r) {
    size_t mid = inst->len / 2, per = 0, extra = 0;
    size_t nt = try_dispatch(mid, 1024, &per, &extra);
    if (nt < 1) {
        fold_chunk(inst->Eq, inst->len, r, 0, mid);
        for (int k = 0; k < inst->arity; k++) fold_chunk(inst->X[k], inst->len, r, 0, mid);
    } else {
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t i = 0; i < nt; i++) {
            size_t s, e;
            task_range(i, per, extra, &s, &e);
            fold_chunk(inst->Eq, inst->len, r, s, e);
            for (int k = 0; k < inst->arity; k++) fold_chunk(inst->X[k], inst->len, r, s, e);
        }
    }
    inst->len = mid;
}

/* sumcheck/prover.go:193-212 + algo.go:209-215 */
static void dispatch_eq_table(ofr_t *eq, size_t len, const ofr_t *q, int bN, const ofr_t *mult) {
    size_t chunk = 256; /* eqTableChunkSize */
    size_t nb_chunks = len / chunk, per = 0, extra = 0;
    size_t nt = try_dispatch(nb_chunks, 1, &per, &extra);
    if (nt < 1) {
        oracle_folded_eq_table(eq, q, bN, mult);
        return;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < nt; i++) {
        size_t s, e;
        task_range(i, per, extra, &s, &e);
        for (size_t c = s; c < e; c++) oracle_chunk_of_eq_table(eq, c, chunk, q, bN, mult);
    }
}

/* sumcheck/prover.go:102-144.  Returns rnd through *rnd (zero when len(claims) < 1). */
static int make_eq_table(ofr_t *eq, size_t len, int bN, const ofr_t *claims, int nclaims, const ofr_t *qprimes,
                         int nq, ofr_t *rnd) {
    memset(rnd, 0, sizeof(ofr_t));
    if (nclaims != nq && nq > 1) return -1; /* panic :113-115 */
    dispatch_eq_table(eq, len, qprimes, bN, NULL);
    if (nclaims < 1) return 0;
    ofr_t init, mult;
    oracle_mimc_hash(&init, claims, (size_t)nclaims);
    mult = init;
    if (nq > 1) {
        ofr_t *tmp = (ofr_t *)malloc(len * sizeof(ofr_t));
        for (int i = 1; i < nq; i++) {
            dispatch_eq_table(tmp, len, qprimes + (size_t)i * bN, bN, &mult);
#pragma omp parallel for schedule(static) if (len >= 2048)
            for (size_t x = 0; x < len; x++) fr_add(&eq[x], &eq[x], &tmp[x]); /* addInPlace algo.go:219-223 */
            fr_mul(&mult, &mult, &init);
        }
        free(tmp);
    }
    *rnd = init;
    return 0;
}

int oracle_sumcheck_prove(int gate, const ofr_t *ark, int arity, int bN, ofr_t *const *X, const ofr_t *qprimes,
                          int nq, const ofr_t *claims, int nclaims, ofr_t *proof_out, ofr_t *challenges_out,
                          ofr_t *final_out) {
    if (arity < 1 || arity > MAX_ARITY || nq < 1) return -1;
    instance_t inst;
    memset(&inst, 0, sizeof(inst));
    inst.len = (size_t)1 << bN;
    inst.arity = arity;
    inst.gate = gate;
    inst.degree = gate_degree(gate) + 1; /* prover.go:95 */
    if (ark) inst.ark = *ark;
    for (int k = 0; k < arity; k++) inst.X[k] = X[k];
    inst.Eq = (ofr_t *)malloc(inst.len * sizeof(ofr_t));
    ofr_t rnd;
    if (make_eq_table(inst.Eq, inst.len, bN, claims, nclaims, qprimes, nq, &rnd) != 0) {
        free(inst.Eq);
        return -1;
    }
    int nc = inst.degree + 1;
    ofr_t evals[MAX_EVALS];
    for (int k = 0; k < bN; k++) { /* prover.go:70-76 */
        dispatch_partial_evals(&inst, evals);
        oracle_interpolate_on_range(&proof_out[(size_t)k * nc], evals, nc);
        ofr_t r;
        oracle_mimc_hash(&r, &proof_out[(size_t)k * nc], (size_t)nc);
        dispatch_folding(&inst, &r);
        challenges_out[k] = r;
    }
    final_out[0] = inst.Eq[0];
    for (int k = 0; k < arity; k++) final_out[1 + k] = inst.X[k][0];
    free(inst.Eq);
    return 0;
}

int oracle_sumcheck_verify(const ofr_t *claims, int nclaims, const ofr_t *proof, int bN, int nc,
                           ofr_t *challenges_out, ofr_t *final_out, ofr_t *recomb_out) { /* verifier.go:28-65 */
    if (nclaims < 1) return -2; /* reference would index claims[0] of an empty slice: panic */
    ofr_t chal, expected, zero = {{0, 0, 0, 0}};
    oracle_mimc_hash(&chal, claims, (size_t)nclaims);
    oracle_eval_univariate(&expected, claims, nclaims, &chal);
    *recomb_out = chal;
    for (int i = 0; i < bN; i++) {
        const ofr_t *p = &proof[(size_t)i * nc];
        ofr_t a0, a1, r;
        oracle_eval_univariate(&a0, p, nc, &zero);
        oracle_eval_univariate(&a1, p, nc, &ONE);
        fr_add(&a0, &a0, &a1);
        if (!fr_eq(&a0, &expected)) return -1;
        oracle_mimc_hash(&r, p, (size_t)nc);
        challenges_out[i] = r;
        oracle_eval_univariate(&expected, p, nc, &r);
    }
    *final_out = expected;
    return 0;
}

void oracle_evaluation(ofr_t *out, int gate, const ofr_t *ark, const ofr_t *qprimes, int nq, int bN,
                       const ofr_t *claims, int nclaims, const ofr_t *const *X, int arity) { /* instance.go:49-68 */
    size_t len = (size_t)1 << bN;
    ofr_t *eq = (ofr_t *)malloc(len * sizeof(ofr_t)), rnd, res = {{0, 0, 0, 0}}, tmp;
    make_eq_table(eq, len, bN, claims, nclaims, qprimes, nq, &rnd);
    for (size_t n = 0; n < len; n++) {
        const ofr_t *buf[MAX_ARITY];
        for (int k = 0; k < arity; k++) buf[k] = &X[k][n];
        gate_eval_n(gate, arity, ark, &tmp, buf);
        fr_mul(&tmp, &tmp, &eq[n]);
        fr_add(&res, &res, &tmp);
    }
    free(eq);
    *out = res;
}

/* ---- circuits (circuit/circuit.go:11-44), MimcCircuit (examples/mimc.go:10-37), gkr ---------------------- */
typedef struct {
    int n_in, in[ORACLE_MAX_GATE_INPUTS];
    int n_out, *out;
    int gate; /* -1 = input layer */
    ofr_t ark;
} layer_t;

/* BuildCircuit (circuit/circuit.go:28-44) from a flat description; returns NULL on a malformed circuit */
static layer_t *build_circuit(const oracle_layer_desc *d, int n) {
    layer_t *c = (layer_t *)calloc((size_t)n, sizeof(layer_t));
    for (int l = 0; l < n; l++) {
        c[l].gate = d[l].gate;
        c[l].n_in = d[l].gate < 0 ? 0 : d[l].n_in;
        if (c[l].n_in > ORACLE_MAX_GATE_INPUTS) goto bad;
        for (int k = 0; k < ORACLE_MAX_GATE_INPUTS; k++) c[l].in[k] = d[l].in[k];
        c[l].ark = d[l].ark;
        c[l].out = (int *)calloc((size_t)n, sizeof(int));
        for (int k = 0; k < c[l].n_in; k++)
            if (c[l].in[k] < 0 || c[l].in[k] >= l) goto bad;
    }
    for (int l = 0; l < n; l++)
        for (int k = 0; k < c[l].n_in; k++) {
            layer_t *p = &c[c[l].in[k]];
            p->out[p->n_out++] = l;
        }
    for (int l = 0; l < n; l++)
        if (c[l].gate < 0 && c[l].n_out > 1) goto bad; /* :36-41 */
    return c;
bad:
    for (int l = 0; l < n; l++) free(c[l].out);
    free(c);
    return NULL;
}
static void free_circuit(layer_t *c, int n) {
    for (int l = 0; l < n; l++) free(c[l].out);
    free(c);
}

#define MIMC_LAYERS 94
static void mimc_descs(oracle_layer_desc *d) { /* examples/mimc.go:10-37 */
    memset(d, 0, sizeof(oracle_layer_desc) * MIMC_LAYERS);
    d[0].gate = -1;
    d[1].gate = -1;
    d[2].gate = ORACLE_GATE_IDENTITY;
    d[2].n_in = 1;
    d[2].in[0] = 0;
    for (int i = 0; i < 91; i++) {
        oracle_layer_desc *l = &d[i + 3];
        l->gate = ORACLE_GATE_CIPHER;
        l->ark = ARKS[i];
        l->n_in = 2;
        l->in[0] = 2;
        l->in[1] = i == 0 ? 1 : i + 2;
    }
}

size_t oracle_circuit_proof_len(const oracle_layer_desc *d, int n, int bN) { /* hints.go:76-116 */
    layer_t *c = build_circuit(d, n);
    if (!c) return 0;
    size_t sc = 0, cl = 0, qp = 0;
    for (int l = 0; l < n; l++) {
        if (c[l].gate >= 0) sc += (size_t)bN * (gate_degree(c[l].gate) + 2);
        cl += (size_t)c[l].n_out;
        qp += (size_t)bN * c[l].n_out;
    }
    free_circuit(c, n);
    return sc + cl + qp + bN;
}
size_t oracle_mimc_proof_len(int bN) { return (size_t)822 * bN + 183 + (size_t)184 * bN; }

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int out_index(const layer_t *p, int layer) { /* sort.SearchInts on the sorted Out list */
    for (int i = 0; i < p->n_out; i++)
        if (p->out[i] == layer) return i;
    return -1;
}

int oracle_gkr_prove_circuit(const oracle_layer_desc *desc, int NL, int bN, const ofr_t *const *inputs, int n_inputs,
                             const ofr_t *qprime, ofr_t *flat_out, ofr_t *outputs_out, double *prove_seconds) {
    layer_t *c = build_circuit(desc, NL);
    if (!c) return -5;
    size_t n = (size_t)1 << bN;
    /* Assign: circuit/assignment.go:12-32 */
    ofr_t **a = (ofr_t **)calloc((size_t)NL, sizeof(ofr_t *));
    for (int l = 0; l < NL; l++) a[l] = (ofr_t *)malloc(n * sizeof(ofr_t));
    for (int l = 0; l < n_inputs; l++) memcpy(a[l], inputs[l], n * sizeof(ofr_t));
    for (int l = n_inputs; l < NL; l++) {
        const ofr_t *xs[ORACLE_MAX_GATE_INPUTS] = {NULL, NULL, NULL, NULL};
        for (int k = 0; k < c[l].n_in; k++) xs[k] = a[c[l].in[k]];
        size_t blk = 4096;
#pragma omp parallel for schedule(static)
        for (size_t s = 0; s < (n + blk - 1) / blk; s++) { /* Layer.Evaluate circuit/circuit.go:48-64 */
            size_t b = s * blk, e = b + blk < n ? b + blk : n;
            const ofr_t *ys[ORACLE_MAX_GATE_INPUTS];
            for (int k = 0; k < ORACLE_MAX_GATE_INPUTS; k++) ys[k] = xs[k] ? xs[k] + b : NULL;
            oracle_gate_eval_batch(c[l].gate, &c[l].ark, a[l] + b, ys, c[l].n_in, e - b);
        }
    }
    if (outputs_out) memcpy(outputs_out, a[NL - 1], n * sizeof(ofr_t));

    /* Prove: gkr/prover.go:21-91 */
    double t0 = now_s();
    ofr_t **claims = (ofr_t **)calloc((size_t)NL, sizeof(ofr_t *)), **qprimes = (ofr_t **)calloc((size_t)NL, sizeof(ofr_t *)),
          **sc = (ofr_t **)calloc((size_t)NL, sizeof(ofr_t *));
    int *has_claims = (int *)calloc((size_t)NL, sizeof(int));
    for (int l = 0; l < NL; l++) {
        int slots = c[l].n_out > 0 ? c[l].n_out : 1;
        claims[l] = (ofr_t *)calloc((size_t)slots, sizeof(ofr_t));
        qprimes[l] = (ofr_t *)calloc((size_t)slots * (bN > 0 ? bN : 1), sizeof(ofr_t));
    }
    memcpy(qprimes[NL - 1], qprime, (size_t)bN * sizeof(ofr_t));
    int rc = 0;
    for (int layer = NL - 1; layer >= 0 && rc == 0; layer--) {
        if (c[layer].gate < 0) break;
        int arity = c[layer].n_in;
        ofr_t *X[ORACLE_MAX_GATE_INPUTS] = {NULL, NULL, NULL, NULL};
        int owned[ORACLE_MAX_GATE_INPUTS] = {0, 0, 0, 0};
        for (int k = 0; k < arity; k++) { /* InputsOfLayer circuit/assignment.go:35-57 */
            int pos = c[layer].in[k];
            if (c[pos].out[0] == layer) {
                X[k] = a[pos];
            } else {
                X[k] = (ofr_t *)malloc(n * sizeof(ofr_t));
                memcpy(X[k], a[pos], n * sizeof(ofr_t));
                owned[k] = 1;
            }
        }
        int nc = gate_degree(c[layer].gate) + 2;
        sc[layer] = (ofr_t *)calloc((size_t)(bN > 0 ? bN : 1) * nc, sizeof(ofr_t));
        ofr_t *next_q = (ofr_t *)calloc((size_t)(bN > 0 ? bN : 1), sizeof(ofr_t));
        ofr_t final[ORACLE_MAX_GATE_INPUTS + 1];
        int nq = layer == NL - 1 ? 1 : c[layer].n_out;
        int ncl = has_claims[layer] ? c[layer].n_out : 0;
        rc = oracle_sumcheck_prove(c[layer].gate, &c[layer].ark, arity, bN, X, qprimes[layer], nq, claims[layer], ncl,
                                   sc[layer], next_q, final);
        for (int i = 1; i <= arity && rc == 0; i++) { /* updateWithSumcheck :66-90 */
            int inp = c[layer].in[i - 1];
            int w = out_index(&c[inp], layer);
            if (w < 0) { rc = -3; break; }
            has_claims[inp] = 1;
            claims[inp][w] = final[i];
            memcpy(qprimes[inp] + (size_t)w * bN, next_q, (size_t)bN * sizeof(ofr_t));
        }
        free(next_q);
        for (int k = 0; k < arity; k++)
            if (owned[k]) free(X[k]);
    }
    if (prove_seconds) *prove_seconds = now_s() - t0;

    /* GkrProofToVec order: hints.go:236-271 */
    size_t cur = 0;
    if (rc == 0) {
        for (int l = 0; l < NL; l++)
            if (sc[l]) {
                size_t cnt = (size_t)bN * (gate_degree(c[l].gate) + 2);
                memcpy(flat_out + cur, sc[l], cnt * sizeof(ofr_t));
                cur += cnt;
            }
        for (int l = 0; l < NL; l++) {
            memcpy(flat_out + cur, claims[l], (size_t)c[l].n_out * sizeof(ofr_t));
            cur += (size_t)c[l].n_out;
        }
        for (int l = 0; l < NL; l++) {
            size_t slots = l == NL - 1 ? 1 : (size_t)c[l].n_out;
            memcpy(flat_out + cur, qprimes[l], slots * bN * sizeof(ofr_t));
            cur += slots * bN;
        }
        if (cur != oracle_circuit_proof_len(desc, NL, bN)) rc = -4;
    }
    for (int l = 0; l < NL; l++) {
        free(a[l]);
        free(claims[l]);
        free(qprimes[l]);
        free(sc[l]);
    }
    free(a);
    free(claims);
    free(qprimes);
    free(sc);
    free(has_claims);
    free_circuit(c, NL);
    return rc;
}

int oracle_gkr_prove_mimc(int bN, const ofr_t *in0, const ofr_t *in1, const ofr_t *qprime, ofr_t *flat_out,
                          ofr_t *outputs_out, double *prove_seconds) {
    oracle_layer_desc d[MIMC_LAYERS];
    mimc_descs(d);
    const ofr_t *ins[2] = {in0, in1};
    return oracle_gkr_prove_circuit(d, MIMC_LAYERS, bN, ins, 2, qprime, flat_out, outputs_out, prove_seconds);
}

int oracle_gkr_verify_circuit(const oracle_layer_desc *desc, int NL, int bN, const ofr_t *flat, const ofr_t *const *inputs,
                              int n_inputs, const ofr_t *outputs, const ofr_t *qprime) { /* gkr/verifier.go:15-132 */
    layer_t *c = build_circuit(desc, NL);
    if (!c) return -5;
    size_t n = (size_t)1 << bN;
    /* locate the three sections of the flat proof */
    const ofr_t **sc = (const ofr_t **)calloc((size_t)NL, sizeof(ofr_t *)), **claims = (const ofr_t **)calloc((size_t)NL, sizeof(ofr_t *)),
                **qps = (const ofr_t **)calloc((size_t)NL, sizeof(ofr_t *));
    size_t cur = 0;
    int max_out = 1;
    for (int l = 0; l < NL; l++) {
        if (c[l].n_out > max_out) max_out = c[l].n_out;
        if (c[l].gate >= 0) {
            sc[l] = flat + cur;
            cur += (size_t)bN * (gate_degree(c[l].gate) + 2);
        }
    }
    for (int l = 0; l < NL; l++) {
        claims[l] = flat + cur;
        cur += (size_t)c[l].n_out;
    }
    for (int l = 0; l < NL; l++) {
        qps[l] = flat + cur;
        cur += (l == NL - 1 ? 1 : (size_t)c[l].n_out) * bN;
    }
    int rc = 0;
    ofr_t *next_q = (ofr_t *)calloc((size_t)(bN > 0 ? bN : 1), sizeof(ofr_t));
    ofr_t *tmp_evals = (ofr_t *)calloc((size_t)max_out, sizeof(ofr_t));
    ofr_t top_claim;
    if (memcmp(qprime, qps[NL - 1], (size_t)bN * sizeof(ofr_t)) != 0) { rc = -10; goto done; } /* :25-30 */
    oracle_evaluate(&top_claim, outputs, n, qprime, bN); /* :36 */
    for (int layer = NL - 1; layer >= 0 && rc == 0; layer--) {
        if (c[layer].gate < 0) break;
        const ofr_t *cl = layer == NL - 1 ? &top_claim : claims[layer];
        int ncl = layer == NL - 1 ? 1 : c[layer].n_out;
        int nc = gate_degree(c[layer].gate) + 2;
        ofr_t next_claim, recomb;
        if (oracle_sumcheck_verify(cl, ncl, sc[layer], bN, nc, next_q, &next_claim, &recomb) != 0) {
            rc = -20 - layer * 10;
            break;
        }
        const ofr_t *sub[ORACLE_MAX_GATE_INPUTS];
        for (int k = 0; k < c[layer].n_in; k++) { /* testSumcheck :74-93 */
            int inp = c[layer].in[k];
            int r_at = out_index(&c[inp], layer);
            if (memcmp(qps[inp] + (size_t)r_at * bN, next_q, (size_t)bN * sizeof(ofr_t)) != 0) rc = -21 - layer * 10;
            sub[k] = &claims[inp][r_at];
        }
        if (rc) break;
        ofr_t expected, eq_eval;
        gate_eval_n(c[layer].gate, c[layer].n_in, &c[layer].ark, &expected, sub);
        for (int i = 0; i < ncl; i++) oracle_eval_eq(&tmp_evals[i], qps[layer] + (size_t)i * bN, next_q, bN);
        oracle_eval_univariate(&eq_eval, tmp_evals, ncl, &recomb);
        fr_mul(&expected, &expected, &eq_eval);
        if (!fr_eq(&expected, &next_claim)) rc = -22 - layer * 10;
    }
    for (int l = 0; l < n_inputs && rc == 0; l++) { /* testInitialRound :120-132 */
        ofr_t actual;
        oracle_evaluate(&actual, inputs[l], n, qps[l], bN);
        if (!fr_eq(&actual, &claims[l][0])) rc = -30 - l;
    }
done:
    free(next_q);
    free(tmp_evals);
    free(sc);
    free(claims);
    free(qps);
    free_circuit(c, NL);
    return rc;
}

int oracle_gkr_verify_mimc(int bN, const ofr_t *flat, const ofr_t *in0, const ofr_t *in1, const ofr_t *outputs,
                           const ofr_t *qprime) {
    oracle_layer_desc d[MIMC_LAYERS];
    mimc_descs(d);
    const ofr_t *ins[2] = {in0, in1};
    return oracle_gkr_verify_circuit(d, MIMC_LAYERS, bN, flat, ins, 2, outputs, qprime);
}
