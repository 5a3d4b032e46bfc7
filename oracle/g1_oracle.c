/* g1_oracle.c -- CPU oracle of BN254 G1 scalar multiplication and multi-scalar multiplication.  TEST INFRASTRUCTURE ONLY
 * (part of libgkr_oracle.so; nothing under gkr-mimc_amd/ includes, links or calls it).
 *
 * Stands for gnark-crypto's (*G1Jac).MultiExp / BatchScalarMultiplicationG1 as called at the reference's
 * prover/gadget/prove.go:76,91,177,189,202,221.  gnark-crypto is an un-vendored dependency (go.mod:7,
 * v0.6.1-0.20220110145513-493bb1c180d9): PARITY UNPINNED against bytes of the Go binary.  The result of an MSM is a group
 * element with unique affine coordinates; this file computes it the plain way -- every [s_i] P_i by left-to-right
 * double-and-add in Jacobian coordinates (dbl-2009-l, madd-2007-bl), the terms summed, one inversion at the end -- and is
 * itself pinned by tests/test_oracle_ec.py against the big-integer affine arithmetic of oracle/pyoracle_ec.py (curve
 * equation, generator (1, 2), [r] G = infinity, random sums).  Deliberately NOT the product's algorithm (signed-window
 * buckets in extended Jacobian coordinates, g1.hip.h): the two share the field modulus and nothing else.
 *
 * Images: a point is gnark-crypto's G1Affine (X, Y: 4 little-endian u64 Montgomery limbs each, infinity = (0, 0)); a
 * scalar is 4 little-endian u64 limbs of the REGULAR-form value (what the reference passes after FromMont). */
#include "gkr_oracle.h"

#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef struct {
    u64 l[4];
} fp_t;

static const u64 Pm[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 PINV = 0x87d20782e4866389ULL; /* -p^-1 mod 2^64 */
static const fp_t FP_ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};

static int fp_geq_p(const u64 t[4]) {
    for (int i = 3; i >= 0; i--) {
        if (t[i] > Pm[i]) return 1;
        if (t[i] < Pm[i]) return 0;
    }
    return 1;
}
static void fp_sub_p(u64 t[4]) {
    u64 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)t[i] - Pm[i] - b;
        t[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
}
static int fp_is_zero(const fp_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a, b, sizeof *a) == 0; }
static void fp_mul(fp_t *z, const fp_t *x, const fp_t *y) { /* CIOS, canonical result */
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)x->l[j] * y->l[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * PINV;
        c = ((u128)m * Pm[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * Pm[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    memcpy(z->l, t, 32);
    if (t[4] || fp_geq_p(z->l)) fp_sub_p(z->l);
}
static void fp_add(fp_t *z, const fp_t *x, const fp_t *y) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)x->l[i] + y->l[i];
        z->l[i] = (u64)c;
        c >>= 64;
    }
    if (fp_geq_p(z->l)) fp_sub_p(z->l);
}
static void fp_sub(fp_t *z, const fp_t *x, const fp_t *y) {
    u64 b = 0, r[4];
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x->l[i] - y->l[i] - b;
        r[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
    if (b) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r[i] + Pm[i];
            r[i] = (u64)c;
            c >>= 64;
        }
    }
    memcpy(z->l, r, 32);
}
static void fp_inv(fp_t *z, const fp_t *a) { /* a^(p-2) */
    u64 e[4] = {Pm[0] - 2, Pm[1], Pm[2], Pm[3]};
    fp_t r = FP_ONE, base = *a;
    for (int i = 253; i >= 0; i--) {
        fp_mul(&r, &r, &r);
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&r, &r, &base);
    }
    *z = r;
}

typedef struct {
    fp_t x, y, z; /* Jacobian: (X / Z^2, Y / Z^3); infinity: Z = 0 */
} jac_t;

static void jac_set_inf(jac_t *p) { memset(p, 0, sizeof *p); }
static void jac_dbl(jac_t *p) { /* dbl-2009-l (a = 0) */
    if (fp_is_zero(&p->z)) return;
    fp_t a, b, c, d, e, f, t;
    fp_mul(&a, &p->x, &p->x);
    fp_mul(&b, &p->y, &p->y);
    fp_mul(&c, &b, &b);
    fp_add(&t, &p->x, &b);
    fp_mul(&t, &t, &t);
    fp_sub(&t, &t, &a);
    fp_sub(&t, &t, &c);
    fp_add(&d, &t, &t);
    fp_add(&e, &a, &a);
    fp_add(&e, &e, &a);
    fp_mul(&f, &e, &e);
    fp_t x3, y3, z3;
    fp_sub(&x3, &f, &d);
    fp_sub(&x3, &x3, &d);
    fp_mul(&z3, &p->y, &p->z);
    fp_add(&z3, &z3, &z3);
    fp_sub(&t, &d, &x3);
    fp_mul(&y3, &e, &t);
    fp_add(&c, &c, &c);
    fp_add(&c, &c, &c);
    fp_add(&c, &c, &c);
    fp_sub(&y3, &y3, &c);
    p->x = x3, p->y = y3, p->z = z3;
}
/* p += (ax, ay) affine, not infinity */
static void jac_madd(jac_t *p, const fp_t *ax, const fp_t *ay) {
    if (fp_is_zero(&p->z)) {
        p->x = *ax, p->y = *ay, p->z = FP_ONE;
        return;
    }
    fp_t z1z1, u2, s2, h, r, t;
    fp_mul(&z1z1, &p->z, &p->z);
    fp_mul(&u2, ax, &z1z1);
    fp_mul(&s2, ay, &p->z);
    fp_mul(&s2, &s2, &z1z1);
    fp_sub(&h, &u2, &p->x);
    fp_sub(&r, &s2, &p->y);
    if (fp_is_zero(&h)) {
        if (fp_is_zero(&r)) jac_dbl(p);
        else jac_set_inf(p);
        return;
    }
    fp_t hh, hhh, v, x3, y3, z3;
    fp_mul(&hh, &h, &h);
    fp_mul(&hhh, &hh, &h);
    fp_mul(&v, &p->x, &hh);
    fp_mul(&x3, &r, &r);
    fp_sub(&x3, &x3, &hhh);
    fp_sub(&x3, &x3, &v);
    fp_sub(&x3, &x3, &v);
    fp_sub(&t, &v, &x3);
    fp_mul(&y3, &r, &t);
    fp_mul(&t, &p->y, &hhh);
    fp_sub(&y3, &y3, &t);
    fp_mul(&z3, &p->z, &h);
    p->x = x3, p->y = y3, p->z = z3;
}
/* p += q: through q's affine image (one inversion: the oracle is not in a hurry) */
static void jac_to_affine(fp_t *ax, fp_t *ay, const jac_t *p) {
    if (fp_is_zero(&p->z)) {
        memset(ax, 0, sizeof *ax);
        memset(ay, 0, sizeof *ay);
        return;
    }
    fp_t zi, zi2, zi3;
    fp_inv(&zi, &p->z);
    fp_mul(&zi2, &zi, &zi);
    fp_mul(&zi3, &zi2, &zi);
    fp_mul(ax, &p->x, &zi2);
    fp_mul(ay, &p->y, &zi3);
}
static void jac_add(jac_t *p, const jac_t *q) {
    if (fp_is_zero(&q->z)) return;
    fp_t ax, ay;
    jac_to_affine(&ax, &ay, q);
    jac_madd(p, &ax, &ay);
}
static void jac_scalar_mul(jac_t *out, const u64 base[8], const u64 s[4]) {
    fp_t bx, by;
    memcpy(bx.l, base, 32);
    memcpy(by.l, base + 4, 32);
    jac_set_inf(out);
    if (fp_is_zero(&bx) && fp_is_zero(&by)) return;
    for (int i = 255; i >= 0; i--) {
        jac_dbl(out);
        if ((s[i >> 6] >> (i & 63)) & 1) jac_madd(out, &bx, &by);
    }
}

int oracle_g1_on_curve(const uint64_t pt[8]) {
    fp_t x, y, l, r, three;
    memcpy(x.l, pt, 32);
    memcpy(y.l, pt + 4, 32);
    if (fp_is_zero(&x) && fp_is_zero(&y)) return 1;
    if (fp_geq_p(x.l) || fp_geq_p(y.l)) return 0;
    fp_mul(&l, &y, &y);
    fp_mul(&r, &x, &x);
    fp_mul(&r, &r, &x);
    fp_add(&three, &FP_ONE, &FP_ONE);
    fp_add(&three, &three, &FP_ONE);
    fp_add(&r, &r, &three);
    return fp_eq(&l, &r);
}
void oracle_g1_scalar_mul(uint64_t out[8], const uint64_t base[8], const uint64_t scalar[4]) {
    jac_t r;
    fp_t ax, ay;
    jac_scalar_mul(&r, base, scalar);
    jac_to_affine(&ax, &ay, &r);
    memcpy(out, ax.l, 32);
    memcpy(out + 4, ay.l, 32);
}
void oracle_g1_batch_scalar_mul(uint64_t *out, const uint64_t base[8], const uint64_t *scalars, size_t n) {
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) oracle_g1_scalar_mul(out + 8 * i, base, scalars + 4 * i);
}
void oracle_g1_add(uint64_t out[8], const uint64_t a[8], const uint64_t b[8]) {
    jac_t p;
    fp_t ax, ay, bx, by;
    memcpy(ax.l, a, 32);
    memcpy(ay.l, a + 4, 32);
    memcpy(bx.l, b, 32);
    memcpy(by.l, b + 4, 32);
    jac_set_inf(&p);
    if (!(fp_is_zero(&ax) && fp_is_zero(&ay))) jac_madd(&p, &ax, &ay);
    if (!(fp_is_zero(&bx) && fp_is_zero(&by))) jac_madd(&p, &bx, &by);
    jac_to_affine(&ax, &ay, &p);
    memcpy(out, ax.l, 32);
    memcpy(out + 4, ay.l, 32);
}
/* out = sum_i [scalars[i]] points[i] */
void oracle_g1_msm(uint64_t out[8], const uint64_t *points, const uint64_t *scalars, size_t n) {
    jac_t total;
    jac_set_inf(&total);
#pragma omp parallel
    {
        jac_t acc;
        jac_set_inf(&acc);
#pragma omp for schedule(dynamic, 16) nowait
        for (long i = 0; i < (long)n; i++) {
            jac_t t;
            jac_scalar_mul(&t, points + 8 * i, scalars + 4 * i);
            jac_add(&acc, &t);
        }
#pragma omp critical
        jac_add(&total, &acc);
    }
    fp_t ax, ay;
    jac_to_affine(&ax, &ay, &total);
    memcpy(out, ax.l, 32);
    memcpy(out + 4, ay.l, 32);
}
