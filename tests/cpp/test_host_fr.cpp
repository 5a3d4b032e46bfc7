// CPU unit test of the product's host-side scalar code (fr_host.h: Montgomery arithmetic, MiMC
// Fiat-Shamir hash, Lagrange interpolation, eval_eq, limb-split reduction) against the C oracle.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../gkr-mimc_amd/csrc/fr_host.h"
#include "../../oracle/gkr_oracle.h"
using hfr::E;
static unsigned long long st = 88172645463325252ULL;
static unsigned long long rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static E rand_e() { E e; do { for (int i = 0; i < 4; i++) e.l[i] = rnd(); e.l[3] &= 0x3fffffffffffffffULL; } while (!hfr::is_canonical(e)); return e; }
static ofr_t O(const E& e) { ofr_t o; memcpy(&o, &e, 32); return o; }
static bool eq(const E& a, const ofr_t& b) { return memcmp(&a, &b, 32) == 0; }
int main() {
    long bad = 0;
    for (int it = 0; it < 200000; it++) {
        E a = rand_e(), b = rand_e(); ofr_t oa = O(a), ob = O(b), oc;
        oracle_fr_mul(&oc, &oa, &ob); if (!eq(hfr::mul(a, b), oc)) bad++;
        oracle_fr_add(&oc, &oa, &ob); if (!eq(hfr::add(a, b), oc)) bad++;
        oracle_fr_sub(&oc, &oa, &ob); if (!eq(hfr::sub(a, b), oc)) bad++;
    }
    for (int n = 1; n <= 12; n++) {                       // hash of n elements, KAT included through the oracle
        std::vector<E> v(n); std::vector<ofr_t> ov(n);
        for (int i = 0; i < n; i++) { v[i] = rand_e(); ov[i] = O(v[i]); }
        ofr_t oh; oracle_mimc_hash(&oh, ov.data(), n);
        if (!eq(hfr::mimc_hash(v.data(), n), oh)) bad++;
    }
    { E twelve = hfr::from_u64(12); ofr_t o12 = O(twelve), oh; oracle_mimc_hash(&oh, &o12, 1);
      if (!eq(hfr::mimc_hash(&twelve, 1), oh)) bad++; }
    hfr::Lagrange lag;
    for (int n = 1; n <= 12; n++) {
        std::vector<E> v(n), out(n); std::vector<ofr_t> ov(n), oo(n);
        for (int i = 0; i < n; i++) { v[i] = rand_e(); ov[i] = O(v[i]); }
        lag.interpolate(out.data(), v.data(), n);
        oracle_interpolate_on_range(oo.data(), ov.data(), n);
        for (int i = 0; i < n; i++) if (!eq(out[i], oo[i])) bad++;
        E x = rand_e(); ofr_t ox = O(x), oe;
        oracle_eval_univariate(&oe, ov.data(), n, &ox);
        if (!eq(hfr::eval_univariate(v.data(), n, x), oe)) bad++;
        std::vector<E> h(n); std::vector<ofr_t> oh(n);
        for (int i = 0; i < n; i++) { h[i] = rand_e(); oh[i] = O(h[i]); }
        oracle_eval_eq(&oe, ov.data(), oh.data(), n);
        if (!eq(hfr::eval_eq(v.data(), h.data(), n), oe)) bad++;
    }
    for (int it = 0; it < 2000; it++) {                   // limb-split sums: sum of k random elements
        int k = 1 + (int)(rnd() % 5000);
        hfr::u64 lanes[8] = {0};
        ofr_t acc; memset(&acc, 0, 32);
        for (int i = 0; i < k; i++) {
            E e = rand_e(); ofr_t oe = O(e);
            for (int j = 0; j < 4; j++) { lanes[2 * j] += e.l[j] & 0xffffffffULL; lanes[2 * j + 1] += e.l[j] >> 32; }
            oracle_fr_add(&acc, &acc, &oe);
        }
        if (!eq(hfr::reduce_limbsplit(lanes), acc)) bad++;
    }
    { E a = rand_e(); ofr_t oa = O(a), oi; oracle_fr_inverse(&oi, &oa); if (!eq(hfr::pow_q_minus_2(a), oi)) bad++; }
    printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
