// Compiled-code counterpart of the reference's TestGKR (gkr/gkr_test.go:14-78) through the C ABI, the way a
// cgo shim would call it: plain pointers to arrays of 4 x u64 Montgomery limbs, no Python in between.
// tests/ may link oracle/: the oracle is the checker.  For each bN: inputs = RandomFrArray-style tables,
// (1) flat proof and outputs are bit-identical to the oracle's, (2) gkrhip_gkr_verify_mimc accepts, (3) the
// oracle's restated gkr.Verify accepts the GPU proof, (4) a corrupted proof is rejected, (5) Fold / FoldedEqTable
// / sumcheck.Prove entry points agree with the oracle on the same buffers, (6) the flat proof is rebuilt into the
// reference's gkr.Proof{SumcheckProofs, Claims, QPrimes} exactly as integration/go/gkr/prover_gkrhip.go's ProofFromFlat
// does (the inverse of GkrProofToVec, prover/gadget/hints.go:236-271) and every claim of every layer is checked
// against the oracle's Evaluate of that layer's table at the claim's point (gkr/gkr_test.go:35-44), (7) a circuit with
// a registered three-input gate (GMiMC t = 4) through the generic session entry points and gkrhip_gkr_verify.
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

// ---- gkr.Proof as the reference lays it out (gkr/prover.go:14-19), indices into the flat vector -----------------
struct Proof {
    std::vector<std::vector<std::vector<ofr_t>>> SumcheckProofs;   // [layer][round][coeff]
    std::vector<std::vector<ofr_t>> Claims;                        // [layer][slot]
    std::vector<std::vector<std::vector<ofr_t>>> QPrimes;          // [layer][slot][coord]
};
struct CLayer {
    int gate;
    std::vector<int> in, out;
};
// mirror of ProofFromFlat (integration/go/gkr/prover_gkrhip.go)
static Proof proof_from_flat(const std::vector<CLayer>& c, const std::vector<int>& degree, int bN, const std::vector<ofr_t>& flat) {
    const int n = (int)c.size();
    Proof p;
    p.SumcheckProofs.resize(n);
    p.Claims.resize(n);
    p.QPrimes.resize(n);
    size_t cur = 0;
    for (int l = 0; l < n; l++) {
        if (c[l].gate < 0) continue;
        const int nc = degree[l] + 2;
        p.SumcheckProofs[l].resize(bN);
        for (int k = 0; k < bN; k++) {
            p.SumcheckProofs[l][k].assign(flat.begin() + cur, flat.begin() + cur + nc);
            cur += nc;
        }
    }
    for (int l = 0; l < n; l++) {
        p.Claims[l].assign(flat.begin() + cur, flat.begin() + cur + c[l].out.size());
        cur += c[l].out.size();
    }
    for (int l = 0; l < n; l++) {
        const size_t slots = l == n - 1 ? 1 : c[l].out.size();
        p.QPrimes[l].resize(slots);
        for (size_t w = 0; w < slots; w++) {
            p.QPrimes[l][w].assign(flat.begin() + cur, flat.begin() + cur + bN);
            cur += bN;
        }
    }
    CHECK(cur == flat.size(), "flat proof length %zu vs %zu consumed", flat.size(), cur);
    return p;
}
static std::vector<ofr_t> flat_from_proof(const Proof& p) {   // GkrProofToVec order
    std::vector<ofr_t> f;
    for (auto& layer : p.SumcheckProofs) for (auto& rnd : layer) f.insert(f.end(), rnd.begin(), rnd.end());
    for (auto& layer : p.Claims) f.insert(f.end(), layer.begin(), layer.end());
    for (auto& layer : p.QPrimes) for (auto& q : layer) f.insert(f.end(), q.begin(), q.end());
    return f;
}
// Check a rebuilt proof against an assignment computed by the oracle's gates: every claim equals Evaluate(table, point),
// every point handed to an input layer is the challenge vector of the consuming layer's sumcheck (Fiat-Shamir over
// the round coefficients), and re-flattening gives the flat vector back.
static void check_proof_structure(const std::vector<CLayer>& c, const std::vector<int>& degree, int bN,
                                  const std::vector<std::vector<ofr_t>>& a, const std::vector<ofr_t>& qp,
                                  const std::vector<ofr_t>& flat, const char* what) {
    const int n = (int)c.size();
    const size_t N = (size_t)1 << bN;
    const Proof p = proof_from_flat(c, degree, bN, flat);
    const std::vector<ofr_t> again = flat_from_proof(p);
    CHECK(again.size() == flat.size() && memcmp(again.data(), flat.data(), flat.size() * sizeof(ofr_t)) == 0, "%s: re-flattened proof differs", what);
    CHECK(bN == 0 || memcmp(p.QPrimes[n - 1][0].data(), qp.data(), bN * sizeof(ofr_t)) == 0, "%s: initial qPrime", what);
    int checked = 0;
    for (int l = 0; l < n - 1; l++)
        for (size_t w = 0; w < c[l].out.size(); w++) {
            ofr_t v;
            oracle_evaluate(&v, a[l].data(), N, p.QPrimes[l][w].data(), bN);
            CHECK(memcmp(&v, &p.Claims[l][w], sizeof v) == 0, "%s: claim [%d][%zu] != Evaluate", what, l, w);
            // the point is the consuming layer's challenge vector
            const int consumer = c[l].out[w];
            for (int k = 0; k < bN; k++) {
                ofr_t r;
                oracle_mimc_hash(&r, p.SumcheckProofs[consumer][k].data(), p.SumcheckProofs[consumer][k].size());
                CHECK(memcmp(&r, &p.QPrimes[l][w][k], sizeof r) == 0, "%s: point [%d][%zu][%d] is not the challenge", what, l, w, k);
            }
            checked++;
        }
    printf("%s bN=%d: %d claims checked against Evaluate\n", what, bN, checked);
}

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
        if (bN == 3 || bN == 7) {   // (6) gkr.Proof rebuilt from the flat vector; all 183 claims == Evaluate(layer, point)
            std::vector<CLayer> c(94);
            std::vector<int> degree(94, 7);
            c[0].gate = c[1].gate = -1;
            c[2].gate = GKRHIP_GATE_IDENTITY; c[2].in = {0}; degree[2] = 1;
            for (int i = 0; i < 91; i++) { c[i + 3].gate = GKRHIP_GATE_CIPHER; c[i + 3].in = {2, i == 0 ? 1 : i + 2}; }
            for (int l = 0; l < 94; l++) for (int q : c[l].in) c[q].out.push_back(l);
            std::vector<std::vector<ofr_t>> a(94, std::vector<ofr_t>(n));
            a[0] = in0; a[1] = in1; a[2] = in0;
            for (int i = 0; i < 91; i++) {
                ofr_t ark; oracle_get_ark(&ark, i);
                const ofr_t* xs[2] = {a[2].data(), a[i == 0 ? 1 : i + 2].data()};
                oracle_gate_eval_batch(ORACLE_GATE_CIPHER, &ark, a[i + 3].data(), xs, 2, n);
            }
            CHECK(memcmp(a[93].data(), outs.data(), n * sizeof(ofr_t)) == 0, "assignment[93] != outputs");
            check_proof_structure(c, degree, bN, a, qp, flat, "MimcCircuit");
        }
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
    // (7) a circuit with a three-input gate through the generic entry points: GMiMC t = 4 (hash/gmimc.go:16-20,52-65)
    for (int bN : {2, 6}) {
        const size_t n = (size_t)1 << bN;
        const int nl = gkrhip_gmimc_circuit(4, nullptr, 0, nullptr);
        CHECK(nl > 0, "gmimc_circuit: %s", gkrhip_last_error());
        std::vector<gkrhip_layer> layers(nl);
        int imap[8];
        CHECK(gkrhip_gmimc_circuit(4, layers.data(), nl, imap) == nl, "gmimc_circuit fill");
        gkrhip_gate_desc gd;
        OK(gkrhip_gate_lookup(layers[nl - 1].gate, &gd));
        CHECK(gd.n_in == 3 && gd.sum_mask == 7 && gd.power == 1 && layers[nl - 1].n_in == 3, "last layer is not the three-input sum gate");
        int n_in = 0;
        while (layers[n_in].gate < 0) n_in++;
        CHECK(n_in == 6, "GMiMC t=4 circuit has %d inputs", n_in);
        // the same circuit for the oracle
        std::vector<oracle_layer_desc> od(nl);
        std::vector<CLayer> c(nl);
        std::vector<int> degree(nl, 1);
        for (int l = 0; l < nl; l++) {
            memset(&od[l], 0, sizeof od[l]);
            const int g = layers[l].gate;
            od[l].gate = g < 0 ? -1 : g == GKRHIP_GATE_IDENTITY ? ORACLE_GATE_IDENTITY : g == GKRHIP_GATE_CIPHER ? ORACLE_GATE_CIPHER
                                 : g == GKRHIP_GATE_ADD ? ORACLE_GATE_ADD : ORACLE_GATE_SUM;
            od[l].n_in = layers[l].n_in;
            c[l].gate = g;
            if (g == GKRHIP_GATE_CIPHER) degree[l] = 7;
            for (int k = 0; k < layers[l].n_in; k++) { od[l].in[k] = layers[l].in[k]; c[l].in.push_back(layers[l].in[k]); }
            memcpy(od[l].ark.l, layers[l].ark, 32);
        }
        for (int l = 0; l < nl; l++) for (int q : c[l].in) c[q].out.push_back(l);
        std::vector<std::vector<ofr_t>> ins(n_in, std::vector<ofr_t>(n));
        for (int k = 0; k < n_in; k++)
            for (size_t i = 0; i < n; i++) oracle_fr_from_u64(&ins[k][i], 0x2545F4914F6CDD1DULL * (i + 3) + 977 * k + bN);
        std::vector<ofr_t> qp(bN);
        oracle_random_fr_array(qp.data(), bN);
        gkrhip_session* s = nullptr;
        OK(gkrhip_session_create(&s, layers.data(), nl, bN));
        for (int k = 0; k < n_in; k++) OK(gkrhip_session_load_input(s, k, (const uint64_t*)ins[k].data()));
        OK(gkrhip_mimc_session_assign(s));
        const size_t len = gkrhip_session_proof_len(s);
        CHECK(len == oracle_circuit_proof_len(od.data(), nl, bN), "circuit proof length");
        std::vector<ofr_t> flat(len), outs(n), oflat(len), oouts(n);
        OK(gkrhip_mimc_session_prove(s, (const uint64_t*)qp.data(), (uint64_t*)flat.data()));
        OK(gkrhip_mimc_session_outputs(s, (uint64_t*)outs.data()));
        std::vector<const ofr_t*> ip(n_in);
        std::vector<const uint64_t*> ipu(n_in);
        for (int k = 0; k < n_in; k++) { ip[k] = ins[k].data(); ipu[k] = (const uint64_t*)ins[k].data(); }
        CHECK(oracle_gkr_prove_circuit(od.data(), nl, bN, ip.data(), n_in, qp.data(), oflat.data(), oouts.data(), nullptr) == 0, "oracle circuit prove");
        CHECK(memcmp(flat.data(), oflat.data(), len * sizeof(ofr_t)) == 0, "GMiMC t=4 transcript differs at bN=%d", bN);
        CHECK(memcmp(outs.data(), oouts.data(), n * sizeof(ofr_t)) == 0, "GMiMC t=4 outputs differ at bN=%d", bN);
        CHECK(gkrhip_gkr_verify(layers.data(), nl, bN, (const uint64_t*)flat.data(), ipu.data(), n_in, (const uint64_t*)outs.data(),
                                (const uint64_t*)qp.data()) == 0, "gkrhip_gkr_verify rejects: %s", gkrhip_last_error());
        CHECK(oracle_gkr_verify_circuit(od.data(), nl, bN, flat.data(), ip.data(), n_in, outs.data(), qp.data()) == 0, "oracle verifier rejects the GMiMC t=4 proof");
        flat[len / 3].l[1] ^= 2;
        CHECK(gkrhip_gkr_verify(layers.data(), nl, bN, (const uint64_t*)flat.data(), ipu.data(), n_in, (const uint64_t*)outs.data(),
                                (const uint64_t*)qp.data()) > 0, "corrupted GMiMC proof accepted");
        flat[len / 3].l[1] ^= 2;
        // every claim of the rebuilt gkr.Proof against the oracle's assignment
        std::vector<std::vector<ofr_t>> a(nl, std::vector<ofr_t>(n));
        for (int l = 0; l < nl; l++) {
            if (l < n_in) { a[l] = ins[l]; continue; }
            const ofr_t* xs[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int k = 0; k < od[l].n_in; k++) xs[k] = a[od[l].in[k]].data();
            oracle_gate_eval_batch(od[l].gate, &od[l].ark, a[l].data(), xs, od[l].n_in, n);
        }
        check_proof_structure(c, degree, bN, a, qp, flat, "GMiMC t=4");
        gkrhip_mimc_session_destroy(s);
    }
    // Proof groups from a compiled caller (gkrhip_mimc_session_prove_group): three MiMC statements with inputs and points of their own,
    // proven in lock-step in one call -- each transcript is the oracle's for its statement; bad arguments are refused with a message.
    {
        const int bN = 8, G = 3;
        const size_t n = (size_t)1 << bN, len = gkrhip_mimc_proof_len(bN);
        std::vector<std::vector<ofr_t>> in0(G, std::vector<ofr_t>(n)), in1(G, std::vector<ofr_t>(n)), qp(G, std::vector<ofr_t>(bN)),
            flat(G, std::vector<ofr_t>(len)), oflat(G, std::vector<ofr_t>(len)), oouts(G, std::vector<ofr_t>(n));
        gkrhip_session* ss[G];
        const uint64_t* qs[G];
        uint64_t* fs[G];
        int rcs[G] = {-1, -1, -1};
        for (int g = 0; g < G; g++) {
            for (size_t i = 0; i < n; i++) {
                oracle_fr_from_u64(&in0[g][i], 0x9E3779B97F4A7C15ULL * (i + 1) + 31 * g);
                oracle_fr_from_u64(&in1[g][i], 0xC2B2AE3D27D4EB4FULL * (i + 5) + 17 * g);
            }
            for (int k = 0; k < bN; k++) oracle_fr_from_u64(&qp[g][k], 0xD6E8FEB86659FD93ULL * (k + 2) + 101 * g);
            CHECK(oracle_gkr_prove_mimc(bN, in0[g].data(), in1[g].data(), qp[g].data(), oflat[g].data(), oouts[g].data(), nullptr) == 0, "oracle prove (group)");
            OK(gkrhip_mimc_session_create(&ss[g], bN));
            OK(gkrhip_mimc_session_load_inputs(ss[g], (const uint64_t*)in0[g].data(), (const uint64_t*)in1[g].data()));
            OK(gkrhip_mimc_session_assign(ss[g]));
            qs[g] = (const uint64_t*)qp[g].data();
            fs[g] = (uint64_t*)flat[g].data();
        }
        OK(gkrhip_mimc_session_prove_group(G, ss, qs, fs, rcs));
        for (int g = 0; g < G; g++) {
            CHECK(rcs[g] == 0, "group proof %d: code %d", g, rcs[g]);
            CHECK(memcmp(flat[g].data(), oflat[g].data(), len * sizeof(ofr_t)) == 0, "group proof %d differs from the oracle's transcript", g);
        }
        gkrhip_session* twice[2] = {ss[0], ss[0]};
        CHECK(gkrhip_mimc_session_prove_group(2, twice, qs, fs, nullptr) != 0 && strstr(gkrhip_last_error(), "twice"), "a session given twice was accepted");
        CHECK(gkrhip_mimc_session_prove_group(0, ss, qs, fs, nullptr) != 0, "an empty group was accepted");
        CHECK(gkrhip_mimc_session_prove_group(9, ss, qs, fs, nullptr) != 0, "a group of nine was accepted");
        for (int g = 0; g < G; g++) gkrhip_mimc_session_destroy(ss[g]);
    }
    // The hint's own safety net (prover/gadget/hints.go:224-228 `if debug`), as a compiled caller would use it: every sumcheck
    // of gkr.Prove is checked before it is returned and re-run if it does not close (a flipped bit of a device sum: same
    // transcript, one re-run counted); with that check off, verify_after_prove turns the wrong proof into an error.
    {
        const int bN = 9;
        const size_t n = (size_t)1 << bN, len = gkrhip_mimc_proof_len(bN);
        std::vector<ofr_t> in0(n), qp(bN), flat(len), outs(n), oflat(len), oouts(n);
        oracle_random_fr_array(in0.data(), n);
        oracle_random_fr_array(qp.data(), bN);
        CHECK(oracle_gkr_prove_mimc(bN, in0.data(), in0.data(), qp.data(), oflat.data(), oouts.data(), nullptr) == 0, "oracle prove");
        uint64_t fails0 = 0, fails1 = 0, checks = 0;
        OK(gkrhip_profile_counter("layer_check_failures", &fails0));
        OK(gkrhip_set_option("test_corrupt_sum", 1));
        OK(gkrhip_set_option("test_corrupt_skip", 40));
        OK(gkrhip_gkr_prove_mimc(bN, (const uint64_t*)in0.data(), (const uint64_t*)in0.data(), (const uint64_t*)qp.data(),
                                 (uint64_t*)flat.data(), (uint64_t*)outs.data()));
        CHECK(memcmp(flat.data(), oflat.data(), len * sizeof(ofr_t)) == 0, "transcript after a corrupted sum differs");
        OK(gkrhip_profile_counter("layer_check_failures", &fails1));
        OK(gkrhip_profile_counter("layer_checks", &checks));
        CHECK(fails1 == fails0 + 1 && checks >= 92, "layer_check_failures %llu -> %llu, checks %llu", (unsigned long long)fails0,
              (unsigned long long)fails1, (unsigned long long)checks);
        OK(gkrhip_set_option("layer_check", 0));
        OK(gkrhip_set_option("verify_after_prove", 1));
        OK(gkrhip_set_option("test_corrupt_sum", 1));
        const int rc = gkrhip_gkr_prove_mimc(bN, (const uint64_t*)in0.data(), (const uint64_t*)in0.data(), (const uint64_t*)qp.data(),
                                             (uint64_t*)flat.data(), (uint64_t*)outs.data());
        CHECK(rc != 0 && strstr(gkrhip_last_error(), "GKR proof was wrong") != nullptr, "a wrong proof was returned (rc %d, %s)", rc, gkrhip_last_error());
        OK(gkrhip_set_option("layer_check", 1));
        OK(gkrhip_gkr_prove_mimc(bN, (const uint64_t*)in0.data(), (const uint64_t*)in0.data(), (const uint64_t*)qp.data(),
                                 (uint64_t*)flat.data(), (uint64_t*)outs.data()));
        CHECK(memcmp(flat.data(), oflat.data(), len * sizeof(ofr_t)) == 0, "transcript with verify_after_prove differs");
        OK(gkrhip_set_option("verify_after_prove", 0));
        CHECK(gkrhip_set_option("no_such_option", 1) != 0, "unknown option accepted");
    }
    // the gate table refuses what the kernels cannot evaluate
    {
        gkrhip_gate_desc bad_gate;
        memset(&bad_gate, 0, sizeof bad_gate);
        strcpy(bad_gate.id, "MulGate");
        bad_gate.n_in = 2; bad_gate.sum_mask = 3; bad_gate.power = 2;
        int id = -1;
        CHECK(gkrhip_gate_register(&bad_gate, &id) != 0, "power-2 gate accepted");
    }
    // error behaviour: the reference panics on a table whose size is not a power of two >= 2
    std::vector<ofr_t> bad(6);
    ofr_t r; oracle_fr_from_u64(&r, 5);
    CHECK(gkrhip_fold((uint64_t*)bad.data(), 6, r.l) != 0 && strlen(gkrhip_last_error()) > 0, "fold accepted 6 elements");
    gkrhip_shutdown();
    printf("abi-harness fails=%d\n", fails);
    return fails ? 1 : 0;
}
