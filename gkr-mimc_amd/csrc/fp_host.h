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

// Fp2 = Fp[u] / (u^2 + 1): the coordinates of G2 (fptower.E2 {A0, A1})
struct E2 {
    E a0, a1;
    bool operator==(const E2& o) const { return a0 == o.a0 && a1 == o.a1; }
};
// field policies of the curve code below (the host twins of FpF / Fp2F in g1.hip.h)
struct HFp {
    typedef E T;
    static const int LIMBS64 = 4;
    static T zero() { return ZERO; }
    static T one() { return ONE; }
    static T mul(const T& a, const T& b) { return hfp::mul(a, b); }
    static T sqr(const T& a) { return hfp::sqr(a); }
    static T add(const T& a, const T& b) { return hfp::add(a, b); }
    static T sub(const T& a, const T& b) { return hfp::sub(a, b); }
    static T dbl(const T& a) { return hfp::dbl(a); }
    static bool is_zero(const T& a) { return hfp::is_zero(a); }
    static T inv(const T& a) { return hfp::inv(a); }
    static T b_curve() { return add(add(ONE, ONE), ONE); }      // y^2 = x^3 + 3
};
struct HFp2 {
    typedef E2 T;
    static const int LIMBS64 = 8;
    static T zero() { return E2{ZERO, ZERO}; }
    static T one() { return E2{ONE, ZERO}; }
    static T mul(const T& a, const T& b) {
        const E v0 = hfp::mul(a.a0, b.a0), v1 = hfp::mul(a.a1, b.a1);
        const E s = hfp::mul(hfp::add(a.a0, a.a1), hfp::add(b.a0, b.a1));
        return E2{hfp::sub(v0, v1), hfp::sub(hfp::sub(s, v0), v1)};
    }
    static T sqr(const T& a) { return mul(a, a); }
    static T add(const T& a, const T& b) { return E2{hfp::add(a.a0, b.a0), hfp::add(a.a1, b.a1)}; }
    static T sub(const T& a, const T& b) { return E2{hfp::sub(a.a0, b.a0), hfp::sub(a.a1, b.a1)}; }
    static T dbl(const T& a) { return add(a, a); }
    static bool is_zero(const T& a) { return hfp::is_zero(a.a0) && hfp::is_zero(a.a1); }
    static T inv(const T& a) {      // conj(a) / (a0^2 + a1^2)
        const E n = hfp::inv(hfp::add(hfp::sqr(a.a0), hfp::sqr(a.a1)));
        return E2{hfp::mul(a.a0, n), hfp::sub(ZERO, hfp::mul(a.a1, n))};
    }
    static T b_curve() {            // the twist's constant 3 / (9 + u)
        const E three = hfp::add(hfp::add(ONE, ONE), ONE);
        E nine = ZERO;
        for (int i = 0; i < 9; i++) nine = hfp::add(nine, ONE);
        return mul(E2{three, ZERO}, inv(E2{nine, ONE}));
    }
};

template <class F>
struct AffH {
    typename F::T x, y;        // infinity = (0, 0), gnark-crypto's encoding
};
template <class F>
struct XyzzH {
    typename F::T x, y, zz, zzz;      // infinity: zz = 0
};
template <class F>
inline XyzzH<F> xyzz_inf() { return XyzzH<F>{F::zero(), F::zero(), F::zero(), F::zero()}; }
template <class F>
inline bool is_inf(const XyzzH<F>& p) { return F::is_zero(p.zz); }
template <class F>
inline void xyzz_dbl(XyzzH<F>& p) {        // dbl-2008-s-1
    if (is_inf(p)) return;
    const typename F::T u = F::dbl(p.y), v = F::sqr(u), w = F::mul(u, v), s = F::mul(p.x, v);
    const typename F::T xx = F::sqr(p.x), m = F::add(F::dbl(xx), xx);
    const typename F::T x3 = F::sub(F::sqr(m), F::dbl(s));
    p.y = F::sub(F::mul(m, F::sub(s, x3)), F::mul(w, p.y));
    p.x = x3;
    p.zz = F::mul(v, p.zz);
    p.zzz = F::mul(w, p.zzz);
}
template <class F>
inline void xyzz_add(XyzzH<F>& p, const XyzzH<F>& q) {      // add-2008-s with the special cases
    if (is_inf(q)) return;
    if (is_inf(p)) {
        p = q;
        return;
    }
    const typename F::T u1 = F::mul(p.x, q.zz), s1 = F::mul(p.y, q.zzz);
    const typename F::T pp_ = F::sub(F::mul(q.x, p.zz), u1), r = F::sub(F::mul(q.y, p.zzz), s1);
    if (F::is_zero(pp_)) {
        if (F::is_zero(r)) xyzz_dbl(p);
        else p = xyzz_inf<F>();
        return;
    }
    const typename F::T pp = F::sqr(pp_), ppp = F::mul(pp_, pp), qq = F::mul(u1, pp);
    const typename F::T x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    p.y = F::sub(F::mul(r, F::sub(qq, x3)), F::mul(s1, ppp));
    p.x = x3;
    p.zz = F::mul(F::mul(p.zz, q.zz), pp);
    p.zzz = F::mul(F::mul(p.zzz, q.zzz), ppp);
}
template <class F>
inline AffH<F> to_affine(const XyzzH<F>& p) {
    if (is_inf(p)) return AffH<F>{F::zero(), F::zero()};
    const typename F::T i = F::inv(F::mul(p.zz, p.zzz));
    return AffH<F>{F::mul(p.x, F::mul(i, p.zzz)), F::mul(p.y, F::mul(i, p.zz))};
}
template <class F>
inline XyzzH<F> from_affine(const AffH<F>& a) {
    if (F::is_zero(a.x) && F::is_zero(a.y)) return xyzz_inf<F>();
    return XyzzH<F>{a.x, a.y, F::one(), F::one()};
}
// y^2 == x^3 + b (or the infinity encoding)
template <class F>
inline bool on_curve(const AffH<F>& a) {
    if (F::is_zero(a.x) && F::is_zero(a.y)) return true;
    return F::sqr(a.y) == F::add(F::mul(F::sqr(a.x), a.x), F::b_curve());
}
typedef AffH<HFp> Aff;
typedef XyzzH<HFp> XYZZ;

}  // namespace hfp
