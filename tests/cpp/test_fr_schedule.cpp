// CPU unit test of the generated Montgomery column schedule (portable branch of fr_bn254.h) against
// the C oracle (tests/ may link oracle/).  Exercises the carry-out-of-first-product case with
// limbs of 0xFFFFFFFF, boundary values and random values, for canonical and lazy (< 2q) inputs.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define FR_CHECK_SKIPS 1   // every multiply-add whose carry instruction the generator left out is checked for wrap-around
#include "../../gkr-mimc_amd/csrc/fr_bn254.h"
#include "../../oracle/gkr_oracle.h"

static u64 rng_state = 0x9e3779b97f4a7c15ULL;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static const u32 Q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
static bool lt_q(const Fr& a) {
    for (int j = 7; j >= 0; j--) { if (a.v[j] < Q[j]) return true; if (a.v[j] > Q[j]) return false; }
    return false;
}
static Fr canon(Fr a) { while (!lt_q(a)) { u32 br = 0; for (int j = 0; j < 8; j++) a.v[j] = fr_subb(a.v[j], Q[j], br, &br); } return a; }
static Fr gen(int mode) {
    Fr a;
    for (int j = 0; j < 8; j++) {
        u64 r = rnd();
        switch (mode) {
            case 0: a.v[j] = (u32)r; break;
            case 1: a.v[j] = 0xFFFFFFFFu; break;
            case 2: a.v[j] = (r & 1) ? 0xFFFFFFFFu : (u32)(r >> 32); break;
            case 3: a.v[j] = (r & 3) ? 0xFFFFFFFFu : 0u; break;
            default: a.v[j] = 0; break;
        }
    }
    a.v[7] &= 0x3FFFFFFFu;          // < 2^254
    return a;
}
int main() {
    long bad = 0, n = 0;
    Fr qm1; for (int j = 0; j < 8; j++) qm1.v[j] = Q[j]; qm1.v[0] -= 1;
    for (int it = 0; it < 400000; it++) {
        Fr a = canon(gen(it % 5)), b = canon(gen((it / 5) % 5));
        if (it == 0) { a = qm1; b = qm1; }
        if (it == 1) { a = qm1; b = fr_one(); }
        ofr_t oa, ob, oc; memcpy(&oa, &a, 32); memcpy(&ob, &b, 32);
        oracle_fr_mul(&oc, &oa, &ob);
        Fr c = fr_mul(a, b);
        if (memcmp(&c, &oc, 32)) bad++;
        oracle_fr_add(&oc, &oa, &ob); c = fr_add(a, b); if (memcmp(&c, &oc, 32)) bad++;
        oracle_fr_sub(&oc, &oa, &ob); c = fr_sub(a, b); if (memcmp(&c, &oc, 32)) bad++;
        // lazy inputs: (a+q)*(b+q) raw must still reduce to the same canonical product
        if (it % 4 == 0) {
            Fr aq, bq; u32 cy = 0; for (int j = 0; j < 8; j++) aq.v[j] = fr_addc(a.v[j], Q[j], cy, &cy);
            cy = 0; for (int j = 0; j < 8; j++) bq.v[j] = fr_addc(b.v[j], Q[j], cy, &cy);
            oracle_fr_mul(&oc, &oa, &ob);
            c = fr_reduce_once(fr_mont_mul_raw(aq, bq));
            if (memcmp(&c, &oc, 32)) bad++;
        }
        // squaring schedule: the same integer as the general product, for canonical and lazy (a + q < 2q) operands
        {
            Fr s0 = fr_mont_sqr_raw(a), s1 = fr_mont_mul_raw(a, a);
            if (memcmp(&s0, &s1, 32)) bad++;
            Fr aq; u32 cy = 0; for (int j = 0; j < 8; j++) aq.v[j] = fr_addc(a.v[j], Q[j], cy, &cy);
            s0 = fr_mont_sqr_raw(aq); s1 = fr_mont_mul_raw(aq, aq);
            if (memcmp(&s0, &s1, 32)) bad++;
        }
        // x^7
        if (it % 8 == 0) {
            ofr_t t; oracle_fr_mul(&t, &oa, &oa); oracle_fr_mul(&t, &t, &oa); oracle_fr_mul(&t, &t, &t); oracle_fr_mul(&t, &t, &oa);
            c = fr_pow7(a); if (memcmp(&c, &t, 32)) bad++;
        }
        n++;
    }
    // deferred reduction: sum_k a_k*b_k accumulated wide (lazy operands < 2q, carry-corner limbs), reduced once,
    // must equal the sum of the individual Montgomery products
    for (int it = 0; it < 20000; it++) {
        u32 A[FR_WIDE_LIMBS] = {0};
        ofr_t sum; memset(&sum, 0, sizeof sum);
        const int terms = 1 + (int)(rnd() % 64);
        for (int k = 0; k < terms; k++) {
            Fr a = canon(gen((it + k) % 5)), b = canon(gen((it / 5 + k) % 5));
            if (it == 0) { a = qm1; b = qm1; }
            ofr_t oa, ob, oc; memcpy(&oa, &a, 32); memcpy(&ob, &b, 32);
            oracle_fr_mul(&oc, &oa, &ob); oracle_fr_add(&sum, &sum, &oc);
            if ((it + k) & 1) {   // lazy representatives a+q, b+q
                u32 cy = 0; for (int j = 0; j < 8; j++) a.v[j] = fr_addc(a.v[j], Q[j], cy, &cy);
                cy = 0; for (int j = 0; j < 8; j++) b.v[j] = fr_addc(b.v[j], Q[j], cy, &cy);
            }
            fr_mac_wide(A, a, b);
        }
        if (it == 1) {            // saturated accumulator below the 2^542 bound: REDC must still fit 9 limbs
            for (int j = 0; j < 16; j++) A[j] = 0xFFFFFFFFu;
            A[16] = 0x3FFFu;
        }
        u32 out[9]; fr_redc_wide(out, A);
        if (it == 1) { n++; if (out[8] > 0x3FFFFFFFu) bad++; continue; }
        // out (288 bits) mod q == sum
        Fr lo; for (int j = 0; j < 8; j++) lo.v[j] = out[j];
        // value = lo + out[8]*2^256: fold the top limb with the Montgomery constant 2^256 mod q = fr_one()
        ofr_t acc; Fr lc = canon(lo); memcpy(&acc, &lc, 32);
        ofr_t one_m; Fr o1 = fr_one(); memcpy(&one_m, &o1, 32);   // regular value 2^256 mod q, as plain limbs
        // top * (2^256 mod q) mod q by double-and-add on plain residues
        ofr_t addend = one_m, topacc; memset(&topacc, 0, sizeof topacc);
        for (u32 t = out[8]; t; t >>= 1) { if (t & 1) oracle_fr_add(&topacc, &topacc, &addend); oracle_fr_add(&addend, &addend, &addend); }
        oracle_fr_add(&acc, &acc, &topacc);
        if (memcmp(&acc, &sum, 32)) bad++;
        if (terms <= 64) {        // the canonicalisation used by k_eq_expand (value < 16q)
            Fr cn = fr_canon_lt16q(out);
            if (memcmp(&cn, &sum, 32)) bad++;
        }
        n++;
    }
    // multiplication by a launch-wide constant through the split images (ca = c * 2^-128, cb = c): the fold product
    {
        ofr_t two128; memset(&two128, 0, sizeof two128); two128.l[2] = 1;      // the plain integer 2^128 as limbs
        for (int it = 0; it < 200000; it++) {
            Fr a = canon(gen(it % 5)), c = canon(gen((it / 5) % 5)), k0 = canon(gen((it / 25) % 5));
            if (it == 0) { a = qm1; c = qm1; k0 = qm1; }
            ofr_t oa, oc, oca, want; memcpy(&oa, &a, 32); memcpy(&oc, &c, 32);
            oracle_fr_mul(&oca, &oc, &two128);                                  // c * 2^128 / 2^256 = c * 2^-128
            Fr ca; memcpy(&ca, &oca, 32);
            oracle_fr_mul(&want, &oa, &oc);
            Fr raw = fr_mul_const2_raw(a, ca, c);
            // raw < 3q; k0 + raw < 4q reduces to k0 + a*c
            ofr_t ok0, sum; memcpy(&ok0, &k0, 32); oracle_fr_add(&sum, &ok0, &want);
            Fr t; u32 cy = 0; for (int j = 0; j < 8; j++) t.v[j] = fr_addc(k0.v[j], raw.v[j], cy, &cy);
            if (cy) bad++;
            Fr got = fr_reduce_lt4q(t);
            if (memcmp(&got, &sum, 32)) bad++;
            n++;
        }
    }
    // The carry planning of tools/gen_mont_asm.py at the edge of its preconditions: operands whose limbs sit at the
    // bounds the planner assumed (every limb 0xFFFFFFFF; top limbs at the largest value of a number below 3q / below
    // q), and values just below 3q with 0xFFFFFFFF patterns.  Any wrap-around of an untracked multiply-add counts.
    {
        const u32 TOP3Q = 0x912ceb58u, TOPQ = 0x30644e72u, TOP2Q = 0x60c89ce5u;   // floor((3q-1)/2^224), floor((q-1)/2^224), floor((2q-1)/2^224)
        auto pat = [&](int mode, u32 top) {
            Fr x;
            for (int j = 0; j < 7; j++) x.v[j] = mode == 0 ? 0xFFFFFFFFu : mode == 1 ? ((j & 1) ? 0xFFFFFFFFu : 0u) : (u32)rnd() | 0xFFFF0000u;
            x.v[7] = top;
            return x;
        };
        for (int it = 0; it < 3000; it++) {
            const int ma = it % 3, mb = (it / 3) % 3;
            Fr a = pat(ma, 0xFFFFFFFFu), b = pat(mb, 0xFFFFFFFFu);   // fr_mont_mul_raw: no precondition on the limbs
            (void)fr_mont_mul_raw(a, b);
            Fr r0, r1;
            fr_mont_mul2_raw(r0, r1, a, b, b, a);
            Fr a3 = pat(ma, TOP3Q), b3 = pat(mb, TOP3Q);            // fr_mac_wide: a, b < 3q
            u32 A[FR_WIDE_LIMBS];
            for (int j = 0; j < FR_WIDE_LIMBS; j++) A[j] = 0xFFFFFFFFu;
            A[16] = 0;
            fr_mac_wide(A, a3, b3);
            Fr a2 = pat(ma, TOP2Q);                                 // fr_mont_sqr_raw: a < 2q
            Fr sq = fr_mont_sqr_raw(a2), sm = fr_mont_mul_raw(a2, a2);
            if (memcmp(&sq, &sm, 32)) bad++;
            Fr ca = pat(mb, TOPQ), cb = pat((mb + 1) % 3, TOPQ);    // fr_mul_const2_raw: a < 3q, ca, cb < q
            (void)fr_mul_const2_raw(a3, ca, cb);
            n++;
        }
        if (fr_skip_overflows) { printf("untracked multiply-adds wrapped: %ld\n", fr_skip_overflows); bad += fr_skip_overflows; }
    }
    printf("cases=%ld bad=%ld skip_overflows=%ld\n", n, bad, fr_skip_overflows);
    return bad ? 1 : 0;
}
