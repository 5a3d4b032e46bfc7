/* gkr_oracle.h -- CPU restatement of the Consensys/gkr-mimc GKR/sumcheck hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (gkr-mimc_amd/, libgkrhip.so) never links, loads or calls it.
 *
 * Plain C (gcc, unsigned __int128 Montgomery arithmetic, OpenMP over the reference's index chunks).
 * Every function cites the reference file:line it follows (paths relative to the reference
 * checkout).  Elements are gnark-crypto `fr.Element` images: 4 little-endian u64 limbs holding
 * value*2^256 mod q, always fully reduced (gnark-crypto v0.6.1-0.20220110145513-493bb1c180d9,
 * go.mod:7 -- external, not vendored; its published CIOS Montgomery algorithm is restated here).
 *
 * PINNING: checked in tests/test_oracle.py against every known-answer value the reference's tests
 * hold for this path (TestMimcCase hash/hash_test.go:21-27; TestFold poly/multilin_test.go:12-31;
 * TestLagrangeCoefficients poly/lagrange_test.go:10-29; TestUnivariate
 * snark/polynomial/univariate_test.go:39-45), against the independent Python big-int restatement
 * (oracle/pyoracle.py) through tests/golden/, and through the reference's own self-consistency
 * tests restated (prover<->verifier).  The Go reference cannot be built in this image (no Go
 * toolchain; gnark-crypto absent), and it stores no transcript vectors: FULL-TRANSCRIPT PARITY WITH
 * THE GO BINARY IS UNPINNED by stored vectors (see DESIGN.md "Oracle").
 */
#ifndef GKR_ORACLE_H
#define GKR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ofr_t; /* Montgomery limbs, little-endian */

/* circuit.Gate implementations (circuit/gates.go:9-21).  IDENTITY and CIPHER are the reference's (copy.go, cipher.go);
 * the others are build-defined members of the same family, variadic as the interface is (arity = len(Layer.In)):
 * ADD = xs[0]+xs[1]+Ark, SUM = xs[0]+...+xs[arity-1]+Ark (Degree 1), SUM_POW7 = (xs[0]+...+xs[arity-1]+Ark)^7 (Degree 7) */
enum { ORACLE_GATE_IDENTITY = 0, ORACLE_GATE_CIPHER = 1, ORACLE_GATE_ADD = 2, ORACLE_GATE_SUM = 3, ORACLE_GATE_SUM_POW7 = 4 };
#define ORACLE_MAX_GATE_INPUTS 4

/* one layer of a circuit (circuit/circuit.go:11-23): gate = -1 for input layers (which come first) */
typedef struct {
    int gate, n_in, in[ORACLE_MAX_GATE_INPUTS];
    ofr_t ark;
} oracle_layer_desc;

/* fr helpers */
void oracle_fr_from_u64(ofr_t *out, uint64_t v);                 /* fr.Element.SetUint64 */
void oracle_fr_mul_generic(ofr_t *o, const ofr_t *a, const ofr_t *b);
void oracle_bench_fr_mul(long n, int generic, double *ns_dependent, double *ns_independent);
void oracle_fr_mul(ofr_t *out, const ofr_t *a, const ofr_t *b);
void oracle_fr_add(ofr_t *out, const ofr_t *a, const ofr_t *b);
void oracle_fr_sub(ofr_t *out, const ofr_t *a, const ofr_t *b);
void oracle_fr_inverse(ofr_t *out, const ofr_t *a);
void oracle_fr_to_regular(uint64_t out[4], const ofr_t *a);      /* FromMont */
void oracle_fr_from_regular(ofr_t *out, const uint64_t in[4]);   /* ToMont (input < q) */

/* hash/mimc.go, common/common.go */
void oracle_mimc_hash(ofr_t *out, const ofr_t *in, size_t n);
void oracle_mimc_keyed_permutation(ofr_t *out, const ofr_t *x, const ofr_t *key);
void oracle_random_fr_array(ofr_t *out, size_t n);
void oracle_get_ark(ofr_t *out, int i);

/* poly */
void oracle_fold(ofr_t *tbl, size_t len, const ofr_t *r);        /* in place; result in [0,len/2) */
void oracle_evaluate(ofr_t *out, const ofr_t *tbl, size_t len, const ofr_t *coords, int n);
void oracle_eval_eq(ofr_t *out, const ofr_t *q, const ofr_t *h, int n);
void oracle_folded_eq_table(ofr_t *out, const ofr_t *q, int n, const ofr_t *mult_or_null);
void oracle_chunk_of_eq_table(ofr_t *out, size_t chunk_id, size_t chunk_size, const ofr_t *q, int n,
                              const ofr_t *mult_or_null);
void oracle_eval_univariate(ofr_t *out, const ofr_t *coeffs, int n, const ofr_t *x);
void oracle_lagrange_coefficient(ofr_t *out /* domain*domain, row l = L_l */, int domain);
int  oracle_interpolate_on_range(ofr_t *out, const ofr_t *values, int n);

/* circuit/gates */
void oracle_gate_eval_batch(int gate, const ofr_t *ark, ofr_t *res, const ofr_t *const *xs, int arity, size_t n);

/* sumcheck/prover.go:46-90.  X[k] (k<arity) are tables of 2^bN elements and are CONSUMED (folded in
 * place) exactly as the reference does.  qprimes: nq*bN elements; claims: nclaims elements.
 * proof_out: bN*(deg+2) coefficients (deg = gate degree + 1); challenges_out: bN; final_out: arity+1.
 * Returns 0, or -1 on the reference's panics (size mismatch). */
int oracle_sumcheck_prove(int gate, const ofr_t *ark, int arity, int bN, ofr_t *const *X,
                          const ofr_t *qprimes, int nq, const ofr_t *claims, int nclaims,
                          ofr_t *proof_out, ofr_t *challenges_out, ofr_t *final_out);

/* sumcheck/verifier.go:28-56; returns 0 if accepted. challenges_out[bN], final_out, recomb_out. */
int oracle_sumcheck_verify(const ofr_t *claims, int nclaims, const ofr_t *proof, int bN, int ncoeff,
                           ofr_t *challenges_out, ofr_t *final_out, ofr_t *recomb_out);

/* sumcheck/instance.go:49-68 */
void oracle_evaluation(ofr_t *out, int gate, const ofr_t *ark, const ofr_t *qprimes, int nq, int bN,
                       const ofr_t *claims, int nclaims, const ofr_t *const *X, int arity);

/* examples/mimc.go + circuit/assignment.go + gkr/prover.go.
 * size of the flat proof (prover/gadget/hints.go:76-116) for MimcCircuit: 822*bN + 183 + 184*bN. */
size_t oracle_mimc_proof_len(int bN);
/* Assign + Prove on inputs in0,in1 (2^bN each, untouched).  flat_out: oracle_mimc_proof_len(bN)
 * elements in GkrProofToVec order (hints.go:236-271) but kept as Montgomery limbs.
 * outputs_out (optional, 2^bN) receives a[93].  prove_seconds (optional) receives the wall time of
 * gkr.Prove alone (assignment excluded, as gkr/gkr_test.go:99-105). */
int oracle_gkr_prove_mimc(int bN, const ofr_t *in0, const ofr_t *in1, const ofr_t *qprime,
                          ofr_t *flat_out, ofr_t *outputs_out, double *prove_seconds);
/* gkr/verifier.go:15-132 on a flat proof. Returns 0 if accepted, else a negative code. */
int oracle_gkr_verify_mimc(int bN, const ofr_t *flat, const ofr_t *in0, const ofr_t *in1,
                           const ofr_t *outputs, const ofr_t *qprime);

/* the same for any layered circuit over the gates above (inputs: the n_inputs leading input layers) */
size_t oracle_circuit_proof_len(const oracle_layer_desc *layers, int n_layers, int bN);
int oracle_gkr_prove_circuit(const oracle_layer_desc *layers, int n_layers, int bN, const ofr_t *const *inputs,
                             int n_inputs, const ofr_t *qprime, ofr_t *flat_out, ofr_t *outputs_out, double *prove_seconds);
int oracle_gkr_verify_circuit(const oracle_layer_desc *layers, int n_layers, int bN, const ofr_t *flat,
                              const ofr_t *const *inputs, int n_inputs, const ofr_t *outputs, const ofr_t *qprime);

int oracle_num_threads(void);
void oracle_set_num_threads(int n);

/* ---- BN254 G1 (g1_oracle.c): gnark-crypto's MultiExp / BatchScalarMultiplicationG1 as called at
 * prover/gadget/prove.go:76,91,177,189,202,221 -- PARITY UNPINNED (un-vendored dependency), pinned on the big-integer
 * arithmetic of pyoracle_ec.py.  Points: G1Affine images (8 u64, Montgomery, infinity = zeros); scalars: 4 u64, regular form. */
int oracle_g1_on_curve(const uint64_t pt[8]);
void oracle_g1_scalar_mul(uint64_t out[8], const uint64_t base[8], const uint64_t scalar[4]);
void oracle_g1_batch_scalar_mul(uint64_t *out, const uint64_t base[8], const uint64_t *scalars, size_t n);
void oracle_g1_add(uint64_t out[8], const uint64_t a[8], const uint64_t b[8]);
void oracle_g1_msm(uint64_t out[8], const uint64_t *points, const uint64_t *scalars, size_t n);

#ifdef __cplusplus
}
#endif
#endif
