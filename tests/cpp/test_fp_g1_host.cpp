// CPU unit test of (1) the generated Fp column schedules (portable branch of fp_bn254.h: fp_mul, fp_sqr and the lazy-range
// add / sub / canon / is_zero) against the host arithmetic of fp_host.h and against the C oracle's G1 (tests/ may link
// oracle/), with every untracked multiply-add checked for wrap-around at the edge of the < 2p precondition; (2) the host
// curve code of fp_host.h (XYZZ add / double / to_affine: the scalar tail of the MSM) against the oracle's Jacobian code.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define FR_CHECK_SKIPS 1
#include "../../gkr-mimc_amd/csrc/fp_bn254.h"
#include "../../gkr-mimc_amd/csrc/fp_host.h"
#include "../../oracle/gkr_oracle.h"

static u64 rng_state = 0x9e3779b97f4a7c15ULL;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static const u32 P32[8] = FP_P_LIMBS;
static const u32 P2[8] = FP_2P_LIMBS;
static bool lt(const Fp& a, const u32 (&m)[8]) {
    for (int j = 7; j >= 0; j--) { if (a.v[j] < m[j]) return true; if (a.v[j] > m[j]) return false; }
    return false;
}
static Fp below2p(int mode) {       // a value in [0, 2p) with extreme limb patterns
    Fp a;
    for (;;) {
        for (int j = 0; j < 8; j++) {
            u64 r = rnd();
            a.v[j] = mode == 0 ? (u32)r : mode == 1 ? 0xFFFFFFFFu : mode == 2 ? ((r & 1) ? 0xFFFFFFFFu : (u32)(r >> 32)) : ((r & 3) ? 0xFFFFFFFFu : 0u);
        }
        a.v[7] = mode == 1 ? 0x60c89ce4u : (u32)(rnd() % 0x60c89ce6u);      // top limb of a number below 2p
        if (lt(a, P2)) return a;
    }
}
static hfp::E to_host(const Fp& a) {     // canonical host image
    hfp::E e;
    memcpy(e.l, a.v, 32);
    return hfp::canon(e);
}
static bool same(const Fp& dev, const hfp::E& host) {
    const hfp::E d = to_host(dev);
    return d == host && lt(dev, P2);
}
int main() {
    long bad = 0, n = 0;
    for (int it = 0; it < 300000; it++) {
        const Fp a = below2p(it % 4), b = below2p((it / 4) % 4);
        const hfp::E ha = to_host(a), hb = to_host(b);
        if (!same(fp_mul(a, b), hfp::mul(ha, hb))) bad++;
        if (!same(fp_sqr(a), hfp::sqr(ha))) bad++;
        { const Fp s = fp_sqr(a), m = fp_mul(a, a); if (memcmp(&s, &m, 32)) bad++; }      // the same integer, not just the same residue
        if (!same(fp_add(a, b), hfp::add(ha, hb))) bad++;
        if (!same(fp_sub(a, b), hfp::sub(ha, hb))) bad++;
        if (!same(fp_neg(a), hfp::sub(hfp::ZERO, ha))) bad++;
        if (fp_is_zero(a) != hfp::is_zero(ha)) bad++;
        { const Fp c = fp_canon(a); hfp::E e; memcpy(e.l, c.v, 32); if (!(e == ha) || !lt(c, P32)) bad++; }
        n++;
    }
    {   // zero in both encodings
        Fp z = fp_zero(), p; memcpy(p.v, P32, 32);
        if (!fp_is_zero(z) || !fp_is_zero(p) || fp_is_zero(fp_one())) bad++;
        if (!same(fp_sub(p, z), hfp::ZERO) || !same(fp_mul(p, below2p(0)), hfp::ZERO)) bad++;
        hfp::E one = to_host(fp_one()); if (!(one == hfp::ONE)) bad++;
    }
    if (fr_skip_overflows) { printf("untracked multiply-adds wrapped: %ld\n", fr_skip_overflows); bad += fr_skip_overflows; }

    // host curve code against the oracle: random multiples of G added and doubled
    uint64_t G[8];
    memcpy(G, hfp::ONE.l, 32);
    { hfp::E two = hfp::add(hfp::ONE, hfp::ONE); memcpy(G + 4, two.l, 32); }
    if (!oracle_g1_on_curve(G)) bad++;
    for (int it = 0; it < 200; it++) {
        uint64_t s1[4] = {rnd(), rnd(), rnd(), rnd() >> 3}, s2[4] = {rnd(), rnd(), rnd(), rnd() >> 3};
        if (it == 0) { memset(s1, 0, 32); }                       // infinity + Q
        if (it == 1) { memcpy(s2, s1, 32); }                      // P + P -> doubling branch
        uint64_t A[8], B[8], want[8];
        oracle_g1_scalar_mul(A, G, s1);
        oracle_g1_scalar_mul(B, G, s2);
        if (it == 2) { hfp::E ny; memcpy(ny.l, A + 4, 32); ny = hfp::sub(hfp::ZERO, ny); memcpy(B, A, 32); memcpy(B + 4, ny.l, 32); }   // P + (-P)
        oracle_g1_add(want, A, B);
        hfp::Aff a, b;
        memcpy(a.x.l, A, 32); memcpy(a.y.l, A + 4, 32); memcpy(b.x.l, B, 32); memcpy(b.y.l, B + 4, 32);
        if (!hfp::on_curve(a) || !hfp::on_curve(b)) bad++;
        hfp::XYZZ p = hfp::from_affine(a);
        // move p off Z = 1: p = 2a - a ... use (a + b) + b - b style instead: scale by doubling then compare 2a + b
        hfp::xyzz_add(p, hfp::from_affine(b));
        hfp::Aff got = hfp::to_affine(p);
        if (memcmp(got.x.l, want, 32) || memcmp(got.y.l, want + 4, 32)) bad++;
        // (2a + b) with general (non-affine) operands: dbl then add of two projective points
        hfp::XYZZ d = hfp::from_affine(a);
        hfp::xyzz_dbl(d);
        hfp::XYZZ q = p;                  // a + b, projective
        hfp::xyzz_add(q, d);              // 3a + b
        uint64_t w2[8], w3[8];
        oracle_g1_add(w2, A, A);
        oracle_g1_add(w3, want, w2);
        got = hfp::to_affine(q);
        if (memcmp(got.x.l, w3, 32) || memcmp(got.y.l, w3 + 4, 32)) bad++;
        n++;
    }
    // Fp2 (the coordinates of G2): the lazy device-portable ops against the host policy, then the host curve code on the twist
    for (int it = 0; it < 20000; it++) {
        const Fp2 a = {below2p(it % 4), below2p((it / 4) % 4)}, b = {below2p((it / 16) % 4), below2p(it % 3)};
        const hfp::E2 ha = {to_host(a.a0), to_host(a.a1)}, hb = {to_host(b.a0), to_host(b.a1)};
        const Fp2 m = fp2_mul(a, b), q2 = fp2_sqr(a);
        const hfp::E2 hm = hfp::HFp2::mul(ha, hb), hq = hfp::HFp2::sqr(ha);
        if (!same(m.a0, hm.a0) || !same(m.a1, hm.a1) || !same(q2.a0, hq.a0) || !same(q2.a1, hq.a1)) bad++;
        n++;
    }
    {
        // gnark-crypto's g2Gen (regular form, little-endian words) on y^2 = x^3 + 3/(9 + u); 2G + G == G + 2G; results on the twist
        const hfp::E x0 = {{0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull}};
        const hfp::E x1 = {{0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull}};
        const hfp::E y0 = {{0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull}};
        const hfp::E y1 = {{0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull}};
        const hfp::AffH<hfp::HFp2> g2 = {hfp::E2{hfp::mul(x0, hfp::R2), hfp::mul(x1, hfp::R2)}, hfp::E2{hfp::mul(y0, hfp::R2), hfp::mul(y1, hfp::R2)}};
        if (!hfp::on_curve(g2)) bad++;
        hfp::XyzzH<hfp::HFp2> d = hfp::from_affine(g2), t = hfp::from_affine(g2);
        hfp::xyzz_dbl(d);                                    // 2G
        hfp::XyzzH<hfp::HFp2> a3 = d;
        hfp::xyzz_add(a3, hfp::from_affine(g2));             // 2G + G
        hfp::xyzz_add(t, d);                                 // G + 2G
        const hfp::AffH<hfp::HFp2> r1 = hfp::to_affine(a3), r2 = hfp::to_affine(t);
        if (!(r1.x == r2.x) || !(r1.y == r2.y) || !hfp::on_curve(r1)) bad++;
        hfp::XyzzH<hfp::HFp2> z = a3;
        hfp::XyzzH<hfp::HFp2> neg = a3;
        neg.y = hfp::HFp2::sub(hfp::HFp2::zero(), neg.y);
        hfp::xyzz_add(z, neg);                               // P + (-P) = infinity
        if (!hfp::is_inf(z)) bad++;
        n++;
    }
    printf("cases=%ld bad=%ld skip_overflows=%ld\n", n, bad, fr_skip_overflows);
    return bad ? 1 : 0;
}
