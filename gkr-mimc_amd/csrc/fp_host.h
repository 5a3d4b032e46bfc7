// fp_host.h -- host-side BN254 BASE-field and G1 arithmetic (4 x 64-bit Montgomery limbs, canonical) for the scalar tail
// of the multi-scalar multiplication: Horner's rule over the W window sums the device hands back and the conversion to
// affine (host_msm.hip.h) -- a few hundred group operations per MSM.  Formulas and special cases are those of g1.hip.h.
// Product code: never includes or links anything under oracle/.
#pragma once
#include <stdint.h>
#include <string.h>

namespace hfp {

typedef unsigned __int128 u128;
typedef uint64_t u64;

struct E {
    u64 l[4];
    bool operator==(const E& o) const { return memcmp(l, o.l, 32) == 0; }
};
static const u64 P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 PINV = 0x87d20782e4866389ULL;      // -p^-1 mod 2^64
static const E ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
static const E R2 = {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
static const E ZERO = {{0, 0, 0, 0}};

static inline bool geq_p(const u64 t[4]) {
    for (int i = 3; i >= 0; i--)
        if (t[i] != P[i]) return t[i] > P[i];
    return true;
}
static inline void sub_p(u64 t[4]) {
    u64 b = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)t[i] - P[i] - b;
        t[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
}
static inline bool is_zero(const E& x) { return (x.l[0] | x.l[1] | x.l[2] | x.l[3]) == 0; }
// CIOS Montgomery product; inputs below 2^256 with x*y < p * 2^256, result canonical
static inline E mul(const E& x, const E& y) {
    u64 t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)x.l[j] * y.l[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        const u64 t5 = (u64)(c >> 64);
        const u64 m = t[0] * PINV;
        c = (u128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t5 + (u64)(c >> 64);
    }
    E r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r.l)) sub_p(r.l);
    return r;
}
static inline E sqr(const E& x) { return mul(x, x); }
static inline E add(const E& x, const E& y) {
    E r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)x.l[i] + y.l[i];
        r.l[i] = (u64)c;
        c >>= 64;
    }
    if (geq_p(r.l)) sub_p(r.l);
    return r;
}
static inline E sub(const E& x, const E& y) {
    E r;
    u64 b = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)x.l[i] - y.l[i] - b;
        r.l[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
    if (b) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.l[i] + P[i];
            r.l[i] = (u64)c;
            c >>= 64;
        }
    }
    return r;
}
static inline E dbl(const E& x) { return add(x, x); }
static inline E canon(const E& x) {      // a device value in [0, 2p) -> [0, p)
    E r = x;
    if (geq_p(r.l)) sub_p(r.l);
    return r;
}
static inline E inv(const E& a) {        // a^(p-2); 0 -> 0
    u64 e[4] = {P[0] - 2, P[1], P[2], P[3]};
    E r = ONE;
    for (int i = 253; i >= 0; i--) {
        r = sqr(r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
    }
    return r;
}

struct Aff {
    E x, y;        // infinity = (0, 0), gnark-crypto's encoding
};
struct XYZZ {
    E x, y, zz, zzz;      // infinity: zz = 0
};
static inline XYZZ xyzz_inf() { return XYZZ{ZERO, ZERO, ZERO, ZERO}; }
static inline bool is_inf(const XYZZ& p) { return is_zero(p.zz); }
static inline void xyzz_dbl(XYZZ& p) {        // dbl-2008-s-1
    if (is_inf(p)) return;
    const E u = dbl(p.y), v = sqr(u), w = mul(u, v), s = mul(p.x, v);
    const E xx = sqr(p.x), m = add(dbl(xx), xx);
    const E x3 = sub(sqr(m), dbl(s));
    p.y = sub(mul(m, sub(s, x3)), mul(w, p.y));
    p.x = x3;
    p.zz = mul(v, p.zz);
    p.zzz = mul(w, p.zzz);
}
static inline void xyzz_add(XYZZ& p, const XYZZ& q) {      // add-2008-s with the special cases
    if (is_inf(q)) return;
    if (is_inf(p)) {
        p = q;
        return;
    }
    const E u1 = mul(p.x, q.zz), s1 = mul(p.y, q.zzz);
    const E pp_ = sub(mul(q.x, p.zz), u1), r = sub(mul(q.y, p.zzz), s1);
    if (is_zero(pp_)) {
        if (is_zero(r)) xyzz_dbl(p);
        else p = xyzz_inf();
        return;
    }
    const E pp = sqr(pp_), ppp = mul(pp_, pp), qq = mul(u1, pp);
    const E x3 = sub(sub(sqr(r), ppp), dbl(qq));
    p.y = sub(mul(r, sub(qq, x3)), mul(s1, ppp));
    p.x = x3;
    p.zz = mul(mul(p.zz, q.zz), pp);
    p.zzz = mul(mul(p.zzz, q.zzz), ppp);
}
static inline Aff to_affine(const XYZZ& p) {
    if (is_inf(p)) return Aff{ZERO, ZERO};
    const E i = inv(mul(p.zz, p.zzz));
    return Aff{mul(p.x, mul(i, p.zzz)), mul(p.y, mul(i, p.zz))};
}
static inline XYZZ from_affine(const Aff& a) {
    if (is_zero(a.x) && is_zero(a.y)) return xyzz_inf();
    return XYZZ{a.x, a.y, ONE, ONE};
}
// y^2 == x^3 + 3 (or the infinity encoding)
static inline bool on_curve(const Aff& a) {
    if (is_zero(a.x) && is_zero(a.y)) return true;
    const E three = add(add(ONE, ONE), ONE);
    return sqr(a.y) == add(mul(sqr(a.x), a.x), three);
}

}  // namespace hfp
