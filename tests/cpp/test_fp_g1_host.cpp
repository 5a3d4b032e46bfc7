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
    printf("cases=%ld bad=%ld skip_overflows=%ld\n", n, bad, fr_skip_overflows);
    return bad ? 1 : 0;
}
