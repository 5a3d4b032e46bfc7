// The Groth16 MSM entry points of include/gkrhip.h driven from compiled code with plain uint64_t arrays -- what the cgo shim
// integration/go/prover/gadget/msm_gkrhip.go passes (unsafe.Pointer(&points[0]), &scalars[0]) -- against the oracle library:
// (*G1Affine).MultiExp over resident bases and in one call (prover/gadget/prove.go:76,91,189,202,221),
// BatchScalarMultiplicationG1 (:177), several goroutine-like host threads sharing one handle and using handles of their own,
// and the error paths (non-canonical coordinate, too many scalars) read back by code.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/gkrhip.h"
#include "../../oracle/gkr_oracle.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
    int fails = 0;
    if (gkrhip_init(0) != 0) {
        printf("init failed: %s\n", gkrhip_last_error());
        return 2;
    }
    // the generator (1, 2) in Montgomery form: oracle_g1_scalar_mul([1]) of a known image
    uint64_t G[8] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL,
                     0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
    if (!oracle_g1_on_curve(G)) fails++;
    const size_t n = 3000;
    std::vector<uint64_t> k(4 * n), s(4 * n), pts(8 * n), want_pts(8 * n);
    for (size_t i = 0; i < n; i++)
        for (int j = 0; j < 4; j++) {
            k[4 * i + j] = j == 3 ? rnd() >> 4 : rnd();
            s[4 * i + j] = j == 3 ? rnd() >> 4 : rnd();
        }
    memset(&s[4 * 7], 0, 32);                                    // a zero scalar
    // BatchScalarMultiplicationG1
    if (gkrhip_g1_batch_scalar_mul(pts.data(), G, k.data(), n, 0) != 0) { printf("batch: %s\n", gkrhip_last_error()); fails++; }
    oracle_g1_batch_scalar_mul(want_pts.data(), G, k.data(), n);
    if (memcmp(pts.data(), want_pts.data(), 64 * n)) { printf("FAIL batch scalar mul\n"); fails++; }
    // MultiExp, one call and resident bases
    uint64_t want[8], got[8];
    oracle_g1_msm(want, pts.data(), s.data(), n);
    if (gkrhip_msm_g1_once(got, pts.data(), s.data(), n, 0) != 0 || memcmp(got, want, 64)) { printf("FAIL msm_g1_once\n"); fails++; }
    gkrhip_g1_bases* b = nullptr;
    if (gkrhip_g1_bases_create(&b, pts.data(), n) != 0 || gkrhip_g1_bases_len(b) != n) { printf("FAIL bases_create\n"); fails++; }
    if (gkrhip_msm_g1(got, b, s.data(), n, 0) != 0 || memcmp(got, want, 64)) { printf("FAIL msm_g1\n"); fails++; }
    // fixed-base tables of the handle (round 6): the same call, the same point; dropped again for what follows
    if (gkrhip_msm_g1_precompute(b, 0) != 0 || gkrhip_msm_g1(got, b, s.data(), n, 0) != 0 || memcmp(got, want, 64)) { printf("FAIL msm_g1 on fixed-base tables\n"); fails++; }
    if (gkrhip_msm_g1_precompute(b, 23) == 0) { printf("FAIL a window of 23 bits accepted\n"); fails++; }
    if (gkrhip_msm_g1_precompute(b, -1) != 0 || gkrhip_msm_g1(got, b, s.data(), n, 0) != 0 || memcmp(got, want, 64)) { printf("FAIL msm_g1 after dropping the tables\n"); fails++; }
    // four host threads on ONE handle (serialised by the handle) and four on handles of their own (concurrent lanes)
    {
        std::vector<std::thread> th;
        int bad[8] = {0};
        for (int t = 0; t < 8; t++)
            th.emplace_back([&, t] {
                uint64_t r[8];
                const size_t m = n - 100 * (size_t)t;           // prefixes of different lengths
                uint64_t w[8];
                oracle_g1_msm(w, pts.data(), s.data(), m);
                if (t < 4) {
                    for (int rep = 0; rep < 3; rep++)
                        if (gkrhip_msm_g1(r, b, s.data(), m, 0) != 0 || memcmp(r, w, 64)) bad[t]++;
                } else {
                    gkrhip_g1_bases* mine = nullptr;
                    if (gkrhip_g1_bases_create(&mine, pts.data(), m) != 0) { bad[t]++; return; }
                    for (int rep = 0; rep < 3; rep++)
                        if (gkrhip_msm_g1(r, mine, s.data(), m, 0) != 0 || memcmp(r, w, 64)) bad[t]++;
                    gkrhip_g1_bases_destroy(mine);
                }
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < 8; t++)
            if (bad[t]) { printf("FAIL concurrent msm, thread %d: %d\n", t, bad[t]); fails++; }
    }
    // errors come back as codes with their messages
    {
        char buf[256];
        std::vector<uint64_t> badp(pts.begin(), pts.begin() + 16);
        badp[3] = ~0ULL;
        int rc = gkrhip_msm_g1_once(got, badp.data(), s.data(), 2, 0);
        if (rc > -16 || gkrhip_last_error_r(rc, buf, sizeof buf) == 0 || !strstr(buf, "canonical")) { printf("FAIL error path 1: %d %s\n", rc, buf); fails++; }
        rc = gkrhip_msm_g1(got, b, s.data(), n, 0);
        if (rc != 0) fails++;
        std::vector<uint64_t> more(4 * (n + 1), 1);
        rc = gkrhip_msm_g1(got, b, more.data(), n + 1, 0);
        if (rc > -16 || gkrhip_last_error_r(rc, buf, sizeof buf) == 0 || !strstr(buf, "scalars for")) { printf("FAIL error path 2: %d %s\n", rc, buf); fails++; }
    }
    // the scalars shared by several sums (bs1 and Bs of prove.go:189,277; with expanded key vectors also ar of :202): one call,
    // scalars in page-locked memory of gkrhip_host_alloc; every result equals the separate call's
    {
        uint64_t G2[16];
        if (gkrhip_g2_generator(G2) != 0) fails++;
        gkrhip_g2_bases* b2 = nullptr;
        gkrhip_g1_bases* ba = nullptr;
        std::vector<uint64_t> pa(pts);
        for (size_t i = 0; i < n; i += 7) memset(&pa[8 * i], 0, 64);          // pk.InfinityA-like holes
        if (gkrhip_g2_bases_generate(&b2, G2, k.data(), n, 0) != 0 || gkrhip_g1_bases_create(&ba, pa.data(), n) != 0) { printf("FAIL bases for the shared call\n"); fails++; }
        void* pin = nullptr;
        if (gkrhip_host_alloc(&pin, 32 * n) != 0 || !pin) { printf("FAIL host_alloc: %s\n", gkrhip_last_error()); fails++; }
        else {
            memcpy(pin, s.data(), 32 * n);
            const uint64_t* ps = (const uint64_t*)pin;
            uint64_t w1[8], w2[16], wa[8], r1[8], r2[16], o1[16], o2[16];
            oracle_g1_msm(wa, pa.data(), s.data(), n);
            if (gkrhip_msm_g1(w1, b, ps, n, 0) != 0 || memcmp(w1, want, 64)) { printf("FAIL msm_g1 from page-locked scalars\n"); fails++; }
            if (gkrhip_msm_g2(w2, b2, ps, n, 0) != 0) { printf("FAIL msm_g2\n"); fails++; }
            if (gkrhip_msm_g1_g2(r1, r2, b, b2, ps, n, 0) != 0 || memcmp(r1, want, 64) || memcmp(r2, w2, 128)) { printf("FAIL msm_g1_g2\n"); fails++; }
            gkrhip_g1_bases* g1s[2] = {ba, b};
            gkrhip_g2_bases* g2s[1] = {b2};
            if (gkrhip_msm_shared(o1, o2, g1s, 2, g2s, 1, ps, n, 0) != 0 || memcmp(o1, wa, 64) || memcmp(o1 + 8, want, 64) || memcmp(o2, w2, 128)) {
                printf("FAIL msm_shared: %s\n", gkrhip_last_error());
                fails++;
            }
            gkrhip_g1_bases* twice[2] = {b, b};
            char buf[256];
            const int rc = gkrhip_msm_shared(o1, o2, twice, 2, nullptr, 0, ps, n, 0);
            if (rc > -16 || gkrhip_last_error_r(rc, buf, sizeof buf) == 0 || !strstr(buf, "twice")) { printf("FAIL shared error path: %d %s\n", rc, buf); fails++; }
            gkrhip_host_free(pin);
        }
        gkrhip_g1_bases_destroy(ba);
        gkrhip_g2_bases_destroy(b2);
    }
    gkrhip_g1_bases_destroy(b);
    printf("abi-msm fails=%d\n", fails);
    return fails ? 1 : 0;
}
