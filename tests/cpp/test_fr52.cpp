// CPU check of the FP64 (5 x 52-bit limbs) Montgomery multiplication experiment tools/fr52.h against the C oracle:
// f52_mont_mul(a, b) == a*b / 2^260 mod q.  Round-toward-zero, as the device kernel sets with s_setreg.
#include <cfenv>
#include <cstdio>
#include <cstring>
#include "../../tools/fr52.h"
#include "../../oracle/gkr_oracle.h"

static u64 rng_state = 0x9e3779b97f4a7c15ULL;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static const u32 Q[8] = {FRQ0, FRQ1, FRQ2, FRQ3, FRQ4, FRQ5, FRQ6, FRQ7};
static bool lt_q(const Fr& a) {
    for (int j = 7; j >= 0; j--) { if (a.v[j] < Q[j]) return true; if (a.v[j] > Q[j]) return false; }
    return false;
}
static Fr canon(Fr a) { while (!lt_q(a)) { u32 br = 0; for (int j = 0; j < 8; j++) a.v[j] = fr_subb(a.v[j], Q[j], br, &br); } return a; }

int main() {
    fesetround(FE_TOWARDZERO);
    long bad = 0, n = 0;
    ofr_t two252; memset(&two252, 0, sizeof two252); two252.l[3] = 1ULL << 60;   // x * 2^252 / 2^256 = x / 16
    for (int it = 0; it < 200000; it++) {
        Fr a, b;
        for (int j = 0; j < 8; j++) {
            const u64 r = rnd();
            a.v[j] = (it % 3 == 0) ? 0xFFFFFFFFu : (u32)r;
            b.v[j] = (it % 5 == 0) ? 0xFFFFFFFFu : (u32)(r >> 32);
        }
        a.v[7] &= 0x3FFFFFFFu; b.v[7] &= 0x3FFFFFFFu;
        a = canon(a); b = canon(b);
        u64 out[F52_LIMBS];
        volatile double guard = 1.0;   // keep the compiler from folding across the rounding-mode change
        (void)guard;
        f52_mont_mul(out, f52_from_fr(a), f52_from_fr(b));
        Fr got = canon(f52_limbs_to_fr(out));
        ofr_t oa, ob, oc, want; memcpy(&oa, &a, 32); memcpy(&ob, &b, 32);
        oracle_fr_mul(&oc, &oa, &ob);           // a*b / 2^256
        oracle_fr_mul(&want, &oc, &two252);     // ... / 2^4
        if (memcmp(&got, &want, 32)) bad++;
        n++;
    }
    printf("cases=%ld bad=%ld\n", n, bad);
    return bad ? 1 : 0;
}
