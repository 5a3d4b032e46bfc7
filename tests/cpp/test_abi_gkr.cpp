// Compiled-code counterpart of the reference's TestGKR (gkr/gkr_test.go:14-78) through the C ABI, the way a
// cgo shim would call it: plain pointers to arrays of 4 x u64 Montgomery limbs, no Python in between.
// tests/ may link oracle/: the oracle is the checker.  For each bN: inputs = RandomFrArray-style tables,
// (1) flat proof and outputs are bit-identical to the oracle's, (2) gkrhip_gkr_verify_mimc accepts, (3) the
// oracle's restated gkr.Verify accepts the GPU proof, (4) a corrupted proof is rejected, (5) Fold / FoldedEqTable
// / sumcheck.Prove entry points agree with the oracle on the same buffers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/gkrhip.h"
#include "../../oracle/gkr_oracle.h"

static int fails = 0;
#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            fails++;                             \
            printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            printf(__VA_ARGS__);                 \
            printf("\n");                        \
        }                                        \
    } while (0)
#define OK(call) CHECK((call) == 0, "%s -> %s", #call, gkrhip_last_error())

int main(int argc, char** argv) {
    if (gkrhip_init(0) != 0) { printf("gkrhip_init failed: %s\n", gkrhip_last_error()); return 2; }
    const int sizes[] = {0, 1, 3, 7, 11};
    for (int bN : sizes) {
        const size_t n = (size_t)1 << bN;
        std::vector<ofr_t> in0(n), in1(n), qp(bN ? bN : 1);
        oracle_random_fr_array(in0.data(), n);                       // common/common.go:49-55
        for (size_t i = 0; i < n; i++) oracle_fr_from_u64(&in1[i], 0x9e3779b97f4a7c15ULL * (i + 1) + bN);
        oracle_random_fr_array(qp.data(), bN);
        const size_t len = gkrhip_mimc_proof_len(bN);
        CHECK(len == oracle_mimc_proof_len(bN), "proof length %zu", len);
        std::vector<ofr_t> flat(len), outs(n), oflat(len), oouts(n);
        OK(gkrhip_gkr_prove_mimc(bN, (const uint64_t*)in0.data(), (const uint64_t*)in1.data(), (const uint64_t*)qp.data(),
                                 (uint64_t*)flat.data(), (uint64_t*)outs.data()));
        CHECK(oracle_gkr_prove_mimc(bN, in0.data(), in1.data(), qp.data(), oflat.data(), oouts.data(), nullptr) == 0, "oracle prove");
        CHECK(memcmp(flat.data(), oflat.data(), len * sizeof(ofr_t)) == 0, "flat proof differs at bN=%d", bN);
        CHECK(memcmp(outs.data(), oouts.data(), n * sizeof(ofr_t)) == 0, "outputs differ at bN=%d", bN);
        CHECK(gkrhip_gkr_verify_mimc(bN, (const uint64_t*)flat.data(), (const uint64_t*)in0.data(), (const uint64_t*)in1.data(),
                                     (const uint64_t*)outs.data(), (const uint64_t*)qp.data()) == 0, "native verifier rejects bN=%d", bN);
        CHECK(oracle_gkr_verify_mimc(bN, flat.data(), in0.data(), in1.data(), outs.data(), qp.data()) == 0, "oracle verifier rejects bN=%d", bN);
        if (bN > 0) {
            flat[len / 2].l[0] ^= 1;                                  // corruption must be caught
            CHECK(gkrhip_gkr_verify_mimc(bN, (const uint64_t*)flat.data(), (const uint64_t*)in0.data(), (const uint64_t*)in1.data(),
                                         (const uint64_t*)outs.data(), (const uint64_t*)qp.data()) > 0, "corrupted proof accepted bN=%d", bN);
        }
        if (bN >= 1) {
            // (*MultiLin).Fold and poly.FoldedEqTable on the same buffers
            ofr_t r; oracle_mimc_hash(&r, in0.data(), 1);
            std::vector<ofr_t> t(in1), ot(in1);
            OK(gkrhip_fold((uint64_t*)t.data(), n, r.l));
            oracle_fold(ot.data(), n, &r);
            CHECK(memcmp(t.data(), ot.data(), (n / 2) * sizeof(ofr_t)) == 0, "fold differs at bN=%d", bN);
            std::vector<ofr_t> eq(n), oeq(n);
            OK(gkrhip_eq_table((uint64_t*)eq.data(), (const uint64_t*)qp.data(), bN, r.l));
            oracle_folded_eq_table(oeq.data(), qp.data(), bN, &r);
            CHECK(memcmp(eq.data(), oeq.data(), n * sizeof(ofr_t)) == 0, "eq table differs at bN=%d", bN);
            // sumcheck.Prove with the cipher gate (sumcheck/prover_test.go:88-94)
            ofr_t ark; oracle_fr_from_u64(&ark, 145646);
            const ofr_t* X[2] = {in0.data(), in1.data()};
            ofr_t claim;
            oracle_evaluation(&claim, ORACLE_GATE_CIPHER, &ark, qp.data(), 1, bN, nullptr, 0, X, 2);
            std::vector<ofr_t> proof(bN * 9), chal(bN), fin(3), oproof(bN * 9), ochal(bN), ofin(3);
            const uint64_t* Xp[2] = {(const uint64_t*)in0.data(), (const uint64_t*)in1.data()};
            OK(gkrhip_sumcheck_prove(GKRHIP_GATE_CIPHER, ark.l, 2, bN, Xp, (const uint64_t*)qp.data(), 1, claim.l, 1,
                                     (uint64_t*)proof.data(), (uint64_t*)chal.data(), (uint64_t*)fin.data()));
            std::vector<ofr_t> c0(in0), c1(in1);                      // the oracle consumes its tables, as the reference does
            ofr_t* Xo[2] = {c0.data(), c1.data()};
            CHECK(oracle_sumcheck_prove(ORACLE_GATE_CIPHER, &ark, 2, bN, Xo, qp.data(), 1, &claim, 1, oproof.data(), ochal.data(), ofin.data()) == 0, "oracle sumcheck");
            CHECK(memcmp(proof.data(), oproof.data(), proof.size() * sizeof(ofr_t)) == 0, "sumcheck proof differs at bN=%d", bN);
            CHECK(memcmp(chal.data(), ochal.data(), chal.size() * sizeof(ofr_t)) == 0, "challenges differ at bN=%d", bN);
            CHECK(memcmp(fin.data(), ofin.data(), 3 * sizeof(ofr_t)) == 0, "final claims differ at bN=%d", bN);
        }
    }
    // error behaviour: the reference panics on a table whose size is not a power of two >= 2
    std::vector<ofr_t> bad(6);
    ofr_t r; oracle_fr_from_u64(&r, 5);
    CHECK(gkrhip_fold((uint64_t*)bad.data(), 6, r.l) != 0 && strlen(gkrhip_last_error()) > 0, "fold accepted 6 elements");
    gkrhip_shutdown();
    printf("abi-harness fails=%d\n", fails);
    return fails ? 1 : 0;
}
