/* gkrhip.h -- C ABI of libgkrhip.so: the MI355X (gfx950) GKR/sumcheck prover for batched MiMC7/BN254.
 *
 * Drop-in boundary for the hot path of Consensys/gkr-mimc (citations are file:line in that
 * repository).  The reference has no FFI layer: the path sits behind plain Go functions, so each
 * entry point below names the Go function whose body a cgo shim replaces (see INTEGRATION.md for
 * the shim).  Every `uint64_t*` field-element array is the memory image of a Go `[]fr.Element`
 * (gnark-crypto, [4]uint64 little-endian Montgomery limbs, canonical), so `unsafe.Pointer(&s[0])`
 * passes with no conversion on the Go side.  Host input buffers are read-only to the library unless
 * stated; outputs are written into caller-allocated buffers.
 *
 * All functions return 0 on success and a non-zero code otherwise.  Every failure returns a code of its own (<= -16);
 * gkrhip_last_error_r(code, buf, cap) returns the message of THAT failure from any thread (a goroutine may have moved
 * to another OS thread between the cgo call and the question), gkrhip_last_error() the last failure of the calling
 * THREAD (the reference panics on the prover side, sumcheck/prover.go:54,114; the Go shim turns non-zero into panic).  Calls block until the result
 * is in host memory.  One context per process; every call works on a lane of its own (stream, hand-off buffers), so
 * calls from different host threads run concurrently (the reference's Prove is called from one goroutine and blocks,
 * sumcheck/prover.go:46-90; concurrent callers there share one worker pool).
 *
 * There is NO CPU fallback: every table-sized operation runs in HIP kernels on the selected GPU and
 * gkrhip_init() fails loudly when no gfx950 device is usable.  The host only performs the
 * inherently serial scalar work (Fiat-Shamir MiMC hashing of <= 91 elements, 9x9 interpolation).
 */
#ifndef GKRHIP_H
#define GKRHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GKRHIP_GATE_IDENTITY 0 /* circuit/gates/copy.go:9-32   : xs[0],              Degree 1 */
#define GKRHIP_GATE_CIPHER 1   /* circuit/gates/cipher.go:11-70: (xs[0]+xs[1]+Ark)^7, Degree 7 */
#define GKRHIP_GATE_ADD 2      /* build-defined (no such circuit.Gate in the reference): xs[0]+xs[1]+Ark, Degree 1;
                                * the non-S-box branches of a GMiMC round (hash/gmimc.go:52-58) */
#define GKRHIP_MAX_GATE_INPUTS 4

/* ---- circuit.Gate plug point (circuit/gates.go:9-21) ----------------------------------------------------------
 * The reference's Gate is an interface with variadic inputs (arity = len(Layer.In)) implemented in Go.  Its native
 * counterpart is a table of gate DESCRIPTORS that the kernels interpret -- no recompilation for a new gate of the
 * family
 *     Eval(xs...) = (sum of the inputs selected by sum_mask + Ark)^power,   power = 1 or 7,  1..4 inputs,
 * with Degree() = power and the layer's Ark supplied per layer (circuit.Layer carries the gate instance; here
 * gkrhip_layer carries the Ark).  The three built-in ids above are entries 0..2 of the same table
 * (identity: sum_mask 1, power 1; cipher: two inputs, power 7; add: two inputs, power 1).  A gate outside the
 * family (e.g. a product of inputs) needs a kernel of its own: gkrhip_gate_register refuses it. */
typedef struct {
    char id[32];             /* Gate.ID() (circuit/gates.go:11): unique name */
    int n_in;                /* number of inputs, 1..GKRHIP_MAX_GATE_INPUTS */
    unsigned int sum_mask;   /* bit k set: input k enters the sum (IdentityGate over [L, R] returns xs[0]: mask 1) */
    int power;               /* 1 or 7 */
} gkrhip_gate_desc;
/* Adds a descriptor and returns its gate id in *gate_id (>= 3); registering an identical descriptor again returns
 * the same id.  Thread-safe. */
int gkrhip_gate_register(const gkrhip_gate_desc *desc, int *gate_id);
int gkrhip_gate_lookup(int gate_id, gkrhip_gate_desc *desc_out);   /* 0 if gate_id is registered */

/* ---- lifecycle ------------------------------------------------------------------------------ */
int gkrhip_init(int device_ordinal);      /* idempotent; selects the GPU, creates the stream/arena */
void gkrhip_shutdown(void);
int gkrhip_device_count(void);
const char *gkrhip_last_error(void);                               /* last failure of the calling thread */
size_t gkrhip_last_error_r(int code, char *buf, size_t cap);        /* the failure that returned `code`, from any thread; returns its length */
const char *gkrhip_version(void);
/* SHA-256 (hex) of the sources and compiler flags this binary was built from; the loaders compare it with the sources
 * they sit next to (gkr-mimc_amd/build.py) so that a stale binary is never called through a changed ABI */
const char *gkrhip_build_id(void);
int gkrhip_device_synchronize(void);      /* waits for every lane's stream */
int gkrhip_mem_info(size_t *free_bytes, size_t *total_bytes);
/* Page-locked host memory for vectors handed over on every call (the scalars of an MSM, the a / b / c of computeH, the inputs of a
 * proof): uploads from these buffers are plain DMA transfers instead of staged copies of pageable memory.  Optional: every
 * entry point accepts ordinary memory. */
int gkrhip_host_alloc(void **out, size_t bytes);
void gkrhip_host_free(void *p);
/* Every call leases a lane (stream, hand-off buffers) from a pool that grows on demand; creating one takes ~3 ms.  A host that will
 * issue n calls at once (the goroutines of ComputeGroth16Proof, several gkr.Prove) can create the lanes ahead (n <= 16).  Optional. */
int gkrhip_reserve_lanes(int n);
/* tuning knobs (measurement only; every setting yields the same transcript): "fold_grid", "fold_split",
 * "g_max", "lat_mode", "wide_mode", "wt_late_lj", "claim_trick", "host_tail", "prelaunch", "prelaunch_lg", "lookahead",
 * "coop", "coop_lg", "coop_wgs", "spec", "spec_lg", "ahead", "solo_boost", "pyr_split" -- applied to every existing lane,
 * waiting for the proofs in flight on them ("Runtime switches", lists the environment variables read at
 * gkrhip_init); "msm_sort_levels" (0: by size, 1 | 2: the one- / two-level counting sort of the MSM forced; same sums);
 * "group_size" (process-wide, default 3; 0 | 1: never): gkrhip_mimc_session_prove calls that meet are proven in groups of this many,
 * "group_wait_us" (default 10000): how long the first caller of such a group waits for the others;
 * "wait_spin_us" (how host threads wait for a round kernel: -2 by the CPUs available -- the default --, -1 always spin, n: spin n us, then sleep).
 * Integrity (process-wide): "layer_check" (default 1) -- every sumcheck the library produces is held against the verifier's own
 * identities before it is returned (sumcheck/verifier.go:41-47 per round, the closing identity of gkr/verifier.go:93-114; host
 * scalar work, microseconds) and run once more in safe mode if it does not close; a second failure is an error, never a
 * wrong proof.  "verify_after_prove" (default 0) -- the one-shot calls (gkrhip_gkr_prove_mimc{,_regular}, gkrhip_gkr_prove) run
 * gkr.Verify on their proof before returning it, as the reference's hint does in debug builds (prover/gadget/hints.go:224-228).
 * "arena_check" (default 0; tests) -- every table handed back to the device arena asks its lane's streams whether they are
 * idle; a release with work still queued is counted ("arena_busy_releases") and its call site named once on stderr; 2: the
 * released table is also filled with 0xff behind the lane's queued work.
 * Fault injection for the tests, each firing once: "test_fail_after_prelaunch", "test_drop_challenge", "test_corrupt_sum" = k
 * (flip one bit of a device sum of round k; "test_corrupt_times" = n afterwards: n times instead of once; "test_corrupt_skip" = j: in the (j+1)-th sumcheck that reaches round k), "test_corrupt_tail" = 1 (flip one bit of the table entries handed to the host). */
int gkrhip_set_option(const char *key, long value);

/* ---- poly.MultiLin (poly/multilin.go) -------------------------------------------------------- */
/* (*MultiLin).Fold(r), poly/multilin.go:19-23: in place on `table` (n elements, n a power of two >= 2);
 * the folded table is the first n/2 elements (the Go shim re-slices to [:mid]). */
int gkrhip_fold(uint64_t *table, size_t n, const uint64_t r[4]);
/* MultiLin.Evaluate(coordinates), poly/multilin.go:59-66 (table untouched). */
int gkrhip_evaluate(uint64_t out[4], const uint64_t *table, size_t n, const uint64_t *coords, int ncoords);
/* poly.FoldedEqTable(preallocated, qPrime, multiplier...), poly/eq.go:41-59; mult may be NULL. Also the
 * result of any sequence of poly.ChunkOfEqTable calls covering the table (poly/eq.go:62-89). */
int gkrhip_eq_table(uint64_t *out, const uint64_t *q, int bN, const uint64_t *mult_or_null);
/* poly.ChunkOfEqTable(preallocatedEq, chunkID, chunkSize, qPrime, multiplier...), poly/eq.go:61-89: fills
 * table[chunk_id*chunk_size, (chunk_id+1)*chunk_size) of the 2^bN-element table; chunk_size a power of two. */
int gkrhip_chunk_of_eq_table(uint64_t *table, size_t chunk_id, size_t chunk_size, const uint64_t *q, int bN,
                             const uint64_t *mult_or_null);

/* ---- circuit.Gate (circuit/gates.go:9-21) ---------------------------------------------------- */
/* Gate.EvalBatch(res, xs...) / Layer.Evaluate, circuit/circuit.go:48-64. */
int gkrhip_gate_eval_batch(int gate, const uint64_t *ark_or_null, uint64_t *res, const uint64_t *const *xs,
                           int arity, size_t n);

/* ---- sumcheck.Prove (sumcheck/prover.go:46-90) ------------------------------------------------ */
/* gate: a GKRHIP_GATE_* or registered id; arity must be the gate's number of inputs (IDENTITY also takes two
 * tables and returns xs[0], as the reference's multi-instance tests do, sumcheck/testing.go:28-57).
 * X[k], k < arity: tables of 2^bN elements (NOT modified; the reference consumes them).
 * qprimes: nq*bN elements; claims: nclaims elements (may be 0: top GKR layer).
 * proof: bN*(Degree+2) coefficients, round-major, low->high (poly.InterpolateOnRange order);
 * challenges: bN; final_claims: arity+1 = [Eq[0], X_1[0], ...] after the last fold. */
int gkrhip_sumcheck_prove(int gate, const uint64_t *ark_or_null, int arity, int bN, const uint64_t *const *X,
                          const uint64_t *qprimes, int nq, const uint64_t *claims, int nclaims,
                          uint64_t *proof, uint64_t *challenges, uint64_t *final_claims);

/* ---- gkr.Prove for examples.MimcCircuit (gkr/prover.go:21-47, examples/mimc.go:10-37) --------- */
/* Number of field elements of the flat proof, = GkrProverHint.NbOutputs (prover/gadget/hints.go:76-116):
 * 822*bN + 183 + 184*bN. */
/* sumcheck.Verify(claims, proof) (sumcheck/verifier.go:28-56): proof = bN rounds of `ncoeffs` coefficients (low -> high).
 * Scalar host work (the Fiat-Shamir chain), no GPU needed.  Returns 0 = accepted -- challenges[bN], *final_claim (the
 * alleged evaluation the caller still has to check against the gate, as the reference leaves it) and *recomb_chal (the
 * recombination challenge of the claims) are filled; 1 + i = the check of round i failed (err != nil in the reference:
 * gkrhip_last_error carries its message "at round i verifier eval at 0 + 1 = ... || expected = ..."); < 0 = bad arguments. */
int gkrhip_sumcheck_verify(const uint64_t *claims, int nclaims, const uint64_t *proof, int bN, int ncoeffs,
                           uint64_t *challenges, uint64_t final_claim[4], uint64_t recomb_chal[4]);
size_t gkrhip_mimc_proof_len(int bN);
/* Circuit.Assign(in0,in1) (circuit/assignment.go:12-32) + gkr.Prove, as GkrProverHint.Call does
 * (prover/gadget/hints.go:220-222).  flat: gkrhip_mimc_proof_len(bN) elements in GkrProofToVec order
 * (hints.go:236-271) kept as Montgomery limbs; outputs_or_null: 2^bN elements = assignment[93]. */
int gkrhip_gkr_prove_mimc(int bN, const uint64_t *in0, const uint64_t *in1, const uint64_t *qprime,
                          uint64_t *flat, uint64_t *outputs_or_null);
/* The same with every buffer in REGULAR form (4 little-endian 64-bit words of the value itself, i.e. big.Int.Bits() padded
 * to four words; every value < q): the body of GkrProverHint.Call (prover/gadget/hints.go:197-233) without 2^(bN+1)
 * SetBigInt and 1 006*bN+183 ToBigIntRegular conversions on the Go side -- the Montgomery conversion of the input tables
 * and of the output table rides on the limb-plane transposition the boundary needs anyway, the few proof elements are
 * converted on the host. */
int gkrhip_gkr_prove_mimc_regular(int bN, const uint64_t *in0, const uint64_t *in1, const uint64_t *qprime,
                                  uint64_t *flat, uint64_t *outputs_or_null);

/* Resident session: the assignment stays in HBM, Prove can be repeated (it never mutates the
 * assignment).  This is what the benchmark times (gkr/gkr_test.go:99-105 excludes Assign). */
typedef struct gkrhip_session gkrhip_session;
typedef gkrhip_session gkrhip_mimc_session;   /* a session over examples.MimcCircuit */
int gkrhip_mimc_session_create(gkrhip_mimc_session **out, int bN);
int gkrhip_mimc_session_load_inputs(gkrhip_mimc_session *s, const uint64_t *in0, const uint64_t *in1);
/* inputs = common.RandomFrArray(2^bN) for both (common/common.go:49-55), generated on the device;
 * index_stride/index_offset select the shard i = j*stride + offset (1, 0 for the whole hypercube). */
int gkrhip_mimc_session_synth_inputs(gkrhip_mimc_session *s, uint64_t index_stride, uint64_t index_offset);
int gkrhip_mimc_session_assign(gkrhip_mimc_session *s);
/* gkr.Prove (gkr/prover.go:21-91) on the resident assignment.  Thread-safe: sessions with lanes of their own prove concurrently from
 * different host threads (the reference's goroutine per statement).  When 24 or more host threads are inside this call with
 * un-sharded sessions of 2^18..2^21 entries, calls that arrive within 10 ms of each other (option "group_wait_us") are proven together as a proof group by
 * the first of them -- the others block until their proof is there (see gkrhip_mimc_session_prove_group: same transcripts, the
 * round kernels of the group in one launch; bN = 20, 72 callers: 66 -> 80 M hashes/s).  gkrhip_set_option("group_size", 0) turns
 * that off; counter "coalesced_proofs". */
int gkrhip_mimc_session_prove(gkrhip_mimc_session *s, const uint64_t *qprime, uint64_t *flat);
/* gkr.Prove (gkr/prover.go:21-91) for n sessions of the same shape from one host thread, in lock-step: proof i is, bit for bit,
 * what gkrhip_mimc_session_prove(s[i], qprime[i], flat[i]) returns, but the round kernels of the n proofs go to the GPU as ONE
 * launch (1 <= n <= 8).  For many small proofs in flight: the reference proves them from n goroutines; here one cgo call
 * proves n of them.  Un-sharded sessions, all different.  rcs (may be NULL) receives every proof's code; returns 0 or the first
 * failing proof's code. */
int gkrhip_mimc_session_prove_group(int n, gkrhip_session *const *s, const uint64_t *const *qprime, uint64_t *const *flat, int *rcs);
int gkrhip_mimc_session_outputs(gkrhip_mimc_session *s, uint64_t *outputs);
/* MultiLin.Evaluate of an assignment layer at `coords` on the device (verifier helper, gkr/verifier.go:36,120-132). */
int gkrhip_mimc_session_evaluate_layer(gkrhip_mimc_session *s, int layer, const uint64_t *coords, uint64_t out[4]);
void gkrhip_mimc_session_destroy(gkrhip_mimc_session *s);

/* ---- gkr.Prove on any layered circuit (circuit/circuit.go:11-44) built from the gates above ----------------
 * A layer is described by its gate, its Ark and the layers it reads (circuit.Layer.In); input layers
 * (gate = -1) come first; Out is computed as BuildCircuit does, with its rule that an input layer has at most
 * one consumer (multi-use tables need an explicit identity layer, as examples/mimc.go:20).  The last layer is
 * the output layer.  Every gkrhip_mimc_session_* entry point accepts such a session (load_inputs needs
 * exactly two input layers; synth_inputs fills every input layer); the flat proof is in GkrProofToVec order
 * for that circuit and has gkrhip_session_proof_len elements. */
typedef struct {
    int gate;          /* -1 input layer, else GKRHIP_GATE_* or an id returned by gkrhip_gate_register */
    int n_in;          /* 0 for inputs, else the gate's number of inputs (len(Layer.In)) */
    int in[GKRHIP_MAX_GATE_INPUTS];
    uint64_t ark[4];   /* Montgomery limbs; ignored for IDENTITY and inputs */
} gkrhip_layer;
int gkrhip_session_create(gkrhip_session **out, const gkrhip_layer *layers, int n_layers, int bN);
int gkrhip_session_load_input(gkrhip_session *s, int input_index, const uint64_t *table);
size_t gkrhip_session_proof_len(const gkrhip_session *s);
int gkrhip_session_num_inputs(const gkrhip_session *s);
/* Circuit.Assign (circuit/assignment.go:12-32) + gkr.Prove (gkr/prover.go:21-47) for any such circuit on host tables in
 * one call: inputs[k] = table of input layer k (2^bN elements); flat: gkrhip_session_proof_len elements in GkrProofToVec
 * order; outputs_or_null: the last layer's table. */
int gkrhip_gkr_prove(const gkrhip_layer *layers, int n_layers, int bN, const uint64_t *const *inputs, int n_inputs,
                     const uint64_t *qprime, uint64_t *flat, uint64_t *outputs_or_null);
/* gkr.Verify (gkr/verifier.go:15-59) for any such circuit on host tables: inputs[k] = table of input layer k,
 * outputs = table of the last layer.  0 = accepted, > 0 = rejected, < 0 = error. */
int gkrhip_gkr_verify(const gkrhip_layer *layers, int n_layers, int bN, const uint64_t *flat, const uint64_t *const *inputs,
                      int n_inputs, const uint64_t *outputs, const uint64_t *qprime);
/* The build-defined circuit of one GMiMC (t = 2) compression, out = GMimcT2.UpdateInplace([s0,s1],[b0,b1])[0]
 * (hash/gmimc.go:52-65; BASELINE config 5): input layers 0..3 = s0, s1, b0, b1.  Returns the number of
 * layers (100); fills layers_out when it is not NULL. */
int gkrhip_gmimc_t2_circuit(gkrhip_layer *layers_out, int capacity);
/* The same for t = 2, 4 or 8 (hash/gmimc.go:16-20): output = UpdateInplace(state, block)[0].  A round is an add layer per
 * linear branch and a cipher layer for the S-box branch; the feed-forward state'[0] + state[0] + block[0] is ONE layer of
 * the registered three-input gate "sum3" (sum_mask 7, power 1).  The reference's round never mixes the branches, so
 * state'[0] depends on one initial branch (91 mod t) besides the feed-forward operands: the input layers are exactly
 * the operands that matter and input_map_out (2t ints, may be NULL) names them -- input layer k is state[j] when
 * input_map_out[k] = j < t, block[j - t] otherwise.  Returns the number of layers. */
int gkrhip_gmimc_circuit(int t, gkrhip_layer *layers_out, int capacity, int *input_map_out);
/* The whole sponge GMimcT{t}.Hash(msg) for messages of nblocks * t elements (hash/gmimc.go:29-49: state = 0, one
 * UpdateInplace per block of t elements, the hash is state[0]); nblocks in 1..64.  The state is carried from block to block
 * (from the second block on the feed-forward is a "sum3" layer on every branch); in the first block, where the state is
 * zero, the round layers are the registered one-input gates "addark1" (x + Ark) and "pow7ark1" ((x + Ark)^7).
 * input_map_out (t * nblocks ints, may be NULL): input layer k is msg[input_map_out[k]] (elements the hash does not depend
 * on -- none for these parameters -- would not be inputs).  Returns the number of layers. */
int gkrhip_gmimc_hash_circuit(int t, int nblocks, gkrhip_layer *layers_out, int capacity, int *input_map_out);

/* ---- gkr.Verify (gkr/verifier.go:15-132): native verifier; MultiLin.Evaluate of the output and input tables
 * runs on the device, the rest is scalar host work.  Returns 0 = accepted, > 0 = rejected (code in
 * gkrhip_last_error), < 0 = error.  The session form uses the resident assignment (inputs = layers 0, 1;
 * outputs = layer 93), so it also checks proofs of sizes whose tables never lived on the host. */
int gkrhip_gkr_verify_mimc(int bN, const uint64_t *flat, const uint64_t *in0, const uint64_t *in1,
                           const uint64_t *outputs, const uint64_t *qprime);
int gkrhip_mimc_session_verify(gkrhip_mimc_session *s, const uint64_t *qprime, const uint64_t *flat);

/* ---- wire format of the production caller (prover/gadget/hints.go) -------------------------------------
 * Bulk conversions, in place, between fr.Element's Montgomery limbs and the regular value as 4 little-endian
 * u64 limbs (what ToBigIntRegular / SetBigInt exchange with big.Int; hints.go:136-141,202-205,236-271), and
 * the body of HashHint.Call for a whole batch: out[i] = hash.MimcKeyedPermutation(x[i], key[i])
 * (hints.go:134-145, hash/mimc.go:31-39; the solver calls it once per hash). */
int gkrhip_to_regular(uint64_t *data, size_t n);
int gkrhip_from_regular(uint64_t *data, size_t n);
int gkrhip_mimc_permutation_batch(uint64_t *out, const uint64_t *x, const uint64_t *key, size_t n);

/* ---- multi-GPU: one process per GPU, hypercube sharded on its log2(world) LOWEST index bits ------------
 * Rank g holds T_g[j] = T[j*world + g] of every table (a dense table over the top bN - log2(world)
 * variables), so all local rounds pair (j, j+mid) exactly as on one GPU and no table data crosses xGMI.
 * Per round the ranks all-reduce (RCCL ncclSum over ncclUint64) the limb-split partial sums -- an exact
 * integer sum; a plain sum of packed 4x64-bit limbs would be wrong -- and every rank hashes the same
 * coefficients.  After the local rounds one element per table per rank is gathered and the last
 * log2(world) rounds run redundantly on every rank.  RCCL is dlopen()ed here, not linked.
 * Bootstrap: rank 0 calls gkrhip_comm_unique_id, the 128 bytes are broadcast by the launcher
 * (torch.distributed in bench.py), every rank calls gkrhip_comm_init after gkrhip_init(local GPU).
 * Once installed, sessions take the GLOBAL bN; load/synth inputs and outputs are this rank's shard
 * (synth: index_stride = world, index_offset = rank).  Host-buffer entry points stay un-sharded. */
int gkrhip_comm_unique_id(uint8_t out[128]);
int gkrhip_comm_init(int world, int rank, const uint8_t unique_id[128]);
/* Same protocol with the exchange on the host: the round kernels hand their 576-byte sums to the host as in the
 * un-sharded case and the ranks add them through a POSIX shared-memory segment `name` (rank 0 creates it and
 * unlinks it once every rank has mapped it; use a name no earlier run can have left behind).  For the ranks of ONE
 * node this is the faster transport (no collective kernel queues behind the compute-bound rounds: DESIGN.md section 7);
 * it is also how several ranks time-share one GPU in the tests. */
int gkrhip_comm_init_shm(int world, int rank, const char *name);
/* Several lanes per rank (1..8): lane k owns its own stream, buffers and communicator (unique id k /
 * segment `name`_k) and pairs with lane k of the other ranks, so `nlanes` independent proofs can be in
 * flight at once (session i runs on lane i mod nlanes: create the sessions in the same order on every
 * rank).  One proof's Fiat-Shamir hashing, all-reduce latency and small rounds then overlap another's
 * big rounds. */
int gkrhip_comm_init_lanes(int world, int rank, int nlanes, const uint8_t *unique_ids /* nlanes x 128 */);
int gkrhip_comm_init_shm_lanes(int world, int rank, int nlanes, const char *name);
/* Several lanes over ONE RCCL communicator ("ticker"): a single thread per rank issues back-to-back all-reduces over
 * the concatenation of every lane's slot on one communicator and stream, so the order of collectives is the same on
 * every rank by construction whatever order the lanes' proofs reach their rounds in (no assumption about hardware
 * queues; docs/DESIGN_rounds1-3.md section 6 has the argument).  A lane's exchange completes in the first tick in which every rank
 * contributed to its slot.  This is the multi-lane RCCL transport of bench.py. */
int gkrhip_comm_init_tick(int world, int rank, int nlanes, const uint8_t unique_id[128]);
/* The same ticker with the tick's all-reduce done on the host through a POSIX shared-memory segment `name` (ranks of one
 * node): every line of the ticker's logic -- slots, counts, re-contribution, votes to stop -- with several ranks sharing
 * ONE GPU, which RCCL cannot do; this is how the multi-rank behaviour of the ticker is tested on a single-GPU box.
 * (GKRHIP_TICK_DEVICE_BUF=1 makes gkrhip_comm_init_tick run ncclAllReduce on device staging buffers with copies around it
 * instead of on the host-mapped buffers.) */
int gkrhip_comm_init_tick_shm(int world, int rank, int nlanes, const char *name);
int gkrhip_comm_tick_stats(uint64_t *ticks, uint64_t *idle_ticks);
int gkrhip_comm_destroy(void);
int gkrhip_comm_info(int *world, int *rank);

/* Host-only scalar pieces of the sharded protocol (no GPU needed): the shard weight eq(q_tail, bits(rank))
 * (q_tail[0] <-> most significant rank bit), the reduction of 8 or 9 limb-split u64 lanes to an element,
 * Fiat-Shamir, and the round coefficients from the eight monomial sums. */
int gkrhip_host_shard_seed(uint64_t out[4], const uint64_t *q_tail, int gamma, int rank);
int gkrhip_host_limbsplit_reduce(uint64_t out[4], const uint64_t *lanes, int nlanes);
int gkrhip_host_mimc_hash(uint64_t out[4], const uint64_t *in, size_t n);
int gkrhip_host_cipher_round_coeffs(uint64_t out[36], const uint64_t *M, const uint64_t c[4], const uint64_t qk[4]);
/* The prover's own check of a finished sumcheck, on host data alone (what gkr.Prove / sumcheck.Prove run on every sumcheck
 * before returning it): the round checks P_i(0) + P_i(1) == expected of sumcheck.Verify (sumcheck/verifier.go:41-47) with the
 * given challenges (no hash), then testSumcheck's closing identity Gate.Eval(finalClaims[1:]) * sum_j rho^j EvalEq(q_j, r) ==
 * P_last(r_last) (gkr/verifier.go:93-114) and finalClaims[0] against that eq value.  claims_are_sums != 0: the claims are
 * the sums (inside gkr.Prove); 0: they only feed Fiat-Shamir (sumcheck/prover.go:128) and round 0 is not held against them.
 * *verdict: 0 closes | 1 + i round i | -1 the closing identity | -2 finalClaims[0]. */
int gkrhip_host_sumcheck_closes(int gate, const uint64_t *ark_or_null, int arity, int bN, const uint64_t *qprimes, int nq,
                                const uint64_t *claims, int nclaims, int claims_are_sums, const uint64_t *proof,
                                const uint64_t *challenges, const uint64_t *final_claims, int *verdict);
/* Round 0 queued ahead of its evaluation point, the host's part: M_j = sum_y eq(q_low, y) S_j(y), j = 1..7,
 * from the 7 * 2^t class sums (S_j(y) at element (j - 1) * 2^t + y) and the t coordinates drawn last (q_low[0] <-> the most
 * significant of the t low index bits). */
int gkrhip_host_ahead_contract(uint64_t out[28], const uint64_t *class_sums, const uint64_t *q_low, int t);

/* Host-only self-test of the proof groups' driver (no GPU; the -m "not gpu" tests call it): n proofs on stacks of their own ask
 * for `steps` launches each, a recorder stands in for the GPU.  Proof 1 asks for another grid at step diverge_at (-1: never) and must
 * get a launch of its own there; the last proof returns after leave_after steps (-1: never) and must leave the group without holding
 * the others up.  counts: launches asked for, launches made, the most proofs in one launch; *verdict: 0 when every combined launch held
 * exactly what its proofs asked for, in order. */
int gkrhip_host_group_selftest(int n, int steps, int diverge_at, int leave_after, uint64_t counts[3], int *verdict);

/* ---- computeH: the H part of Groth16's Krs (prover/gadget/prove.go:308-359) --------------------------------------
 * The next prover cost once GKR is fast: three inverse FFTs, three coset FFTs, the pointwise
 * (a*b - c) * (-2)^-1, one inverse coset FFT over BN254 Fr, FromMont.  a, b, c: n Montgomery elements each (zero-padded to
 * the domain by the library); cardinality: a power of two >= n, or 0 for the next power of two (fft.NewDomain's choice);
 * h: `cardinality` REGULAR-form values in the reference's order (the coefficients of H at bit-reversed positions: this
 * version of computeH does not bit-reverse after the last FFTInverse(DIF)).  The transforms are gnark-crypto's
 * fft.Domain (un-vendored dependency, v0.6.1-0.20220110145513-493bb1c180d9): the result is pinned on that package's
 * published algorithm and on the identity H * (X^n - 1) = A*B - C, not on bytes of the Go binary ("parity unpinned").
 * The G1 MSMs of the same function are gkrhip_msm_g1 below. */
int gkrhip_compute_h(uint64_t *h, const uint64_t *a, const uint64_t *b, const uint64_t *c, size_t n, size_t cardinality);
/* computeH on device-resident vectors: *avg_ms = HIP-event time per computeH (no PCIe), *passes = passes over HBM,
 * *bytes = HBM bytes those passes move (32 B read + 32 B written per element of every array a pass names). */
int gkrhip_bench_compute_h(int logn, int warmup, int iters, double *avg_ms, int *passes, double *bytes);


/* ---- G1 multi-scalar multiplication: gnark-crypto's (*G1Jac).MultiExp / (*G1Affine).MultiExp -----------------------
 * as the reference's Groth16 prover calls them: prover/gadget/prove.go:76,91 (krsNotGkr, KrsPrivNotGkr over
 * pk.privKNotGkr), :189 (bs1 over pk.G1.B), :202 (ar over pk.G1.A), :221 (krs2 over pk.G1.Z with the h of computeH).
 * Memory images are gnark-crypto's: a point is a bn254.G1Affine = {X, Y fp.Element} = 8 uint64 (Montgomery form, canonical;
 * the point at infinity is (0, 0)); a scalar is an fr.Element = 4 uint64.  The reference passes scalars in REGULAR form
 * (it calls FromMont first: prove.go:66,116-120,355-357) -- that is the default here; GKRHIP_MSM_SCALARS_MONT says they are
 * still in Montgomery form (MultiExpConfig.ScalarsMont).  The result is written as a G1Affine; a G1Jac caller takes
 * (X, Y, 1) -- any Jacobian representative of the same point is equivalent for every use the reference makes of it
 * (AddMixed, AddAssign, ScalarMultiplication, FromJacobian: prove.go:194-196,207-210,236-262).
 * The bases of these calls are proving-key vectors, fixed across proofs: gkrhip_g1_bases_create uploads them once and
 * they stay in HBM (like a session's assignment); gkrhip_msm_g1 then moves only the scalars.  gkrhip_msm_g1_once is
 * the one-call form with the exact shape of MultiExp(points, scalars, config).
 * gnark-crypto is an un-vendored dependency (v0.6.1-0.20220110145513-493bb1c180d9): the result is a group element whose
 * affine coordinates are unique, pinned by the test oracle's big-integer arithmetic, not by bytes of the Go binary
 * ("parity unpinned").  n <= 2^26.  The G2 MSM of the same function (prove.go:277, Bs over pk.G2.B) is the gkrhip_*_g2 family
 * below: the same kernels over Fp2 coordinates; a bn254.G2Affine is {X, Y fptower.E2} = {X.A0, X.A1, Y.A0, Y.A1} = 16 uint64. */
#define GKRHIP_MSM_SCALARS_MONT 1
typedef struct gkrhip_g1_bases gkrhip_g1_bases;
int gkrhip_g1_bases_create(gkrhip_g1_bases **out, const uint64_t *points /* n x 8 */, size_t n);
/* bases[i] = [scalars[i]] base, computed on the device and left there (a fixed-base batch as in Groth16's setup) */
int gkrhip_g1_bases_generate(gkrhip_g1_bases **out, const uint64_t base[8], const uint64_t *scalars /* n x 4 */, size_t n, int flags);
size_t gkrhip_g1_bases_len(const gkrhip_g1_bases *b);
int gkrhip_g1_bases_read(const gkrhip_g1_bases *b, uint64_t *out /* count x 8 */, size_t first, size_t count);
void gkrhip_g1_bases_destroy(gkrhip_g1_bases *b);
/* out = sum_{i < n} [scalars[i]] bases[i], n <= gkrhip_g1_bases_len(b) */
int gkrhip_msm_g1(uint64_t out_affine[8], gkrhip_g1_bases *b, const uint64_t *scalars /* n x 4 */, size_t n, int flags);
int gkrhip_msm_g1_once(uint64_t out_affine[8], const uint64_t *points, const uint64_t *scalars, size_t n, int flags);
/* window size of the bucket method for this handle: 0 = chosen from n (default), 2..16 forced (every choice gives the same
 * point; the parity tests sweep it) */
int gkrhip_msm_g1_set_window(gkrhip_g1_bases *b, int c);
/* Fixed-base tables (round 6).  The bases of the reference's MultiExp calls are proving-key vectors (pk.G1.A, pk.G1.B, pk.G1.Z,
 * pk.G1.K, pk.G2.B: prove.go:76,91,189,202,221,277), the same for every proof: this computes [2^(o_j)] P_i (o_j: the first bit of window j) for every window once
 * (W = ceil(255 / c) windows of floor(255 / W) bits or one more; W times the handle's points in HBM; seconds for 2^24 points) and the handle's own MSMs -- gkrhip_msm_g1 /
 * gkrhip_msm_g2 -- then sort every window into ONE bucket space of 2^(c-1) buckets: 12 or 13 additions per scalar instead of 16.
 * Same sums (the tests hold both paths against the oracle).  c = 0: chosen from the number of points (22 from 2^22, 20 from 2^17),
 * 8..22 forced, -1 drops the tables.  gkrhip_compute_h_msm_g1 takes them too, and so do the calls that
 * share a sort between handles (gkrhip_msm_g1_g2, gkrhip_msm_shared) when EVERY handle of the call has tables of one window size
 * (the tables' sort once; otherwise the per-window path). */
int gkrhip_msm_g1_precompute(gkrhip_g1_bases *b, int c);
/* bn254.BatchScalarMultiplicationG1(base, scalars) (prove.go:177): out[i] = [scalars[i]] base as G1Affine */
int gkrhip_g1_batch_scalar_mul(uint64_t *out /* n x 8 */, const uint64_t base[8], const uint64_t *scalars, size_t n, int flags);
/* h = computeH(a, b, c, domain) (prove.go:128, 308-359) followed by krs2.MultiExp(pk.G1.Z, h, cfg) (prove.go:221) in one call with H
 * never leaving the device: out = sum_i [h[i]] bases_z[i] over the `cardinality` values of H (the reference passes all of h;
 * gkrhip_g1_bases_len(bases_z) >= cardinality).  h_or_null also returns H itself (regular form, the reference's order). */
int gkrhip_compute_h_msm_g1(uint64_t out_affine[8], gkrhip_g1_bases *bases_z, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                            size_t n, size_t cardinality, uint64_t *h_or_null);
/* G2: (*G2Jac).MultiExp(points, scalars, config) (prove.go:277) and BatchScalarMultiplicationG2; same conventions, 16 uint64 per point */
typedef struct gkrhip_g2_bases gkrhip_g2_bases;
int gkrhip_g2_bases_create(gkrhip_g2_bases **out, const uint64_t *points /* n x 16 */, size_t n);
int gkrhip_g2_bases_generate(gkrhip_g2_bases **out, const uint64_t base[16], const uint64_t *scalars /* n x 4 */, size_t n, int flags);
size_t gkrhip_g2_bases_len(const gkrhip_g2_bases *b);
int gkrhip_g2_bases_read(const gkrhip_g2_bases *b, uint64_t *out /* count x 16 */, size_t first, size_t count);
void gkrhip_g2_bases_destroy(gkrhip_g2_bases *b);
int gkrhip_msm_g2(uint64_t out_affine[16], gkrhip_g2_bases *b, const uint64_t *scalars /* n x 4 */, size_t n, int flags);
int gkrhip_msm_g2_once(uint64_t out_affine[16], const uint64_t *points, const uint64_t *scalars, size_t n, int flags);
/* bs1.MultiExp(pk.G1.B, wireValuesB, cfg) and Bs.MultiExp(pk.G2.B, wireValuesB, cfg) (prove.go:189,277) are over the same scalars:
 * one upload, one decoding and one sort of the scalars serve both sums.  The two handles must hold the same number of points;
 * the G1 handle's window size is used for both. */
int gkrhip_msm_g1_g2(uint64_t out_g1[8], uint64_t out_g2[16], gkrhip_g1_bases *b1, gkrhip_g2_bases *b2,
                     const uint64_t *scalars /* n x 4 */, size_t n, int flags);
/* The general form: k1 G1 sums (out_g1: k1 x 8) and k2 G2 sums (out_g2: k2 x 16) over one scalar vector.  With the proving key's
 * vectors expanded by points at infinity where pk.InfinityA / pk.InfinityB drop wires (prove.go:136-160), ar (:202), bs1 (:189) and
 * Bs (:277) all run over the unfiltered wireValues and share its upload and sort.  A point at infinity (0, 0) adds nothing. */
int gkrhip_msm_shared(uint64_t *out_g1, uint64_t *out_g2, gkrhip_g1_bases *const *g1, size_t k1, gkrhip_g2_bases *const *g2, size_t k2,
                      const uint64_t *scalars /* n x 4 */, size_t n, int flags);
int gkrhip_msm_g2_set_window(gkrhip_g2_bases *b, int c);
int gkrhip_msm_g2_precompute(gkrhip_g2_bases *b, int c);      /* as gkrhip_msm_g1_precompute */
int gkrhip_g2_batch_scalar_mul(uint64_t *out /* n x 16 */, const uint64_t base[16], const uint64_t *scalars, size_t n, int flags);
int gkrhip_g2_generator(uint64_t out[16]);      /* gnark-crypto's g2Gen (bn254.Generators), Montgomery image */
int gkrhip_bench_msm_g2(int logn, int c_or_0, int warmup, int iters, double *avg_ms, double phase_ms[5], int *c_used,
                        double *host_tail_ms, uint64_t result_or_null[16]);
/* MSM of 2^logn synthetic device-resident bases ([k_i] G, k_i pseudo-random) and scalars (pseudo-random below q):
 * *avg_ms = HIP-event time per MSM up to the window sums' arrival on the host (no scalar upload); phase_ms[5] = digit sort,
 * bucket accumulation, big buckets, window reduction, copy of the window sums; *c_used = the window size; result_or_null = the
 * affine sum of the last run (the caller checks it against gkrhip_msm_g1 on the same data read back). */
int gkrhip_bench_msm_g1(int logn, int c_or_0, int warmup, int iters, double *avg_ms, double phase_ms[5], int *c_used,
                        double *host_tail_ms, uint64_t result_or_null[8]);

/* The same on fixed-base tables (gkrhip_msm_g1_precompute with c_or_0): *precompute_ms = the tables' one-time cost (host clock),
 * not part of *avg_ms; phase_ms[0] = digits, radix sort and run boundaries. */
int gkrhip_bench_msm_g1_fixed_base(int logn, int c_or_0, int warmup, int iters, double *avg_ms, double phase_ms[5], int *c_used,
                                   double *host_tail_ms, double *precompute_ms, uint64_t result_or_null[8]);
int gkrhip_bench_msm_g2_fixed_base(int logn, int c_or_0, int warmup, int iters, double *avg_ms, double phase_ms[5], int *c_used,
                                   double *host_tail_ms, double *precompute_ms, uint64_t result_or_null[16]);

/* ---- measurement hooks ------------------------------------------------------------------------ */
/* Device-resident fold micro-benchmark (shape of BenchmarkFolding, poly/multilin_test.go:55-78):
 * ntab tables of n elements (table[i] = Montgomery(i)), r = 5, `iters` timed out-of-place folds after
 * `warmup`; *avg_ms = mean time per fold from HIP events on the library's stream around the `iters` back-to-back folds;
 * *isolated_ms_or_null (optional) = mean over the same number of folds launched one at a time on an idle GPU, each with
 * its own event pair -- the per-kernel duration a profiler reports. */
int gkrhip_bench_fold(size_t n, int ntab, int warmup, int iters, double *avg_ms, double *isolated_ms_or_null);
/* sumcheck.Prove micro-benchmarks on device-resident tables, shaped like the reference's own
 * (sumcheck/prover_test.go:96-125; instances of sumcheck/testing.go:11-57 with L = R = [0, 1, 2, ...]):
 * kind 0 = BenchmarkWithCipherGate (CipherGate, Ark = 145646, one point RandomFrArray(bn)); kind 1 =
 * BenchmarkMultiIdentity (IdentityGate, `ninstance` points q_i[j] = i*j + i with their true claims).  The instance is
 * built outside the timer; *avg_ms = wall-clock per Prove (Fiat-Shamir hashing included); final_claim0 (may be NULL)
 * receives finalClaims[0] of the last run. */
int gkrhip_bench_sumcheck(int kind, int bn, int ninstance, int warmup, int iters, double *avg_ms, uint64_t final_claim0[4]);
/* BenchmarkPartialEvalWithCipher's shape (sumcheck/prover_test.go:127-147): the instance of
 * InitializeCipherGateInstance(bn) with its Eq table built once, then `iters` x dispatchPartialEvals of round 0 (nine
 * evaluations over 2^(bn-1) pairs, reference-shaped evaluator), sums handed to the host every call.  *us_per_call = wall
 * clock per dispatch; evals0 (may be NULL) = evals[0] of the last call. */
int gkrhip_bench_partial_eval(int bn, int warmup, int iters, double *us_per_call, uint64_t evals0[4]);
/* Per-kernel accounting of the calling process since the last reset: HIP-event time of every fold
 * launch whose input table has >= min_n elements. */
int gkrhip_profile_reset(size_t min_n);
int gkrhip_profile_get(uint64_t *fold_launches, double *fold_ms, double *fold_bytes,
                       uint64_t *peval_launches, double *peval_ms, double *peval_modmuls);

/* Host-side wall-clock split of the fused cipher rounds since the last reset: Fiat-Shamir hashing,
 * waiting for the round kernel, launching, other scalar work (all in ms), and the number of rounds. */
int gkrhip_profile_host(uint64_t *rounds, double *hash_ms, double *wait_ms, double *launch_ms, double *other_ms);
/* How often the serial-latency paths ran since the last reset: rounds whose kernel was queued ahead of its challenge
 * (it polls a host-mapped slot), round-0 launches that used products computed during the previous layer, rounds run by
 * the cooperative eight-lanes-per-pair kernel.  The parity tests use it to prove that a switch selected the path. */
int gkrhip_profile_latency(uint64_t *prelaunched_rounds, uint64_t *lookahead_round0, uint64_t *coop_rounds);
/* The same counters by name ("prelaunched_rounds", "lookahead_round0", "coop_rounds", and "spec_rounds": rounds whose
 * sums were computed speculatively for the eight candidate values 0..7 of the previous challenge while the host was still
 * hashing, and interpolated at the true challenge), and "chal_retries": layers whose rounds were run a second time, nothing
 * queued ahead of its challenge, because a waiting kernel's time (1 s) ran out -- the result is the same, the proof is merely
 * late; "layer_checks": sumchecks held against the verifier's identities before they were returned; "layer_check_failures":
 * those that did not close and were run a second time in safe mode (see gkrhip_set_option, "layer_check") -- any value
 * other than 0 outside the fault-injection tests means the device side slipped and deserves a report; "ahead_round0": cipher
 * layers whose round 0 was queued by the layer before them; "hw_queues_set_by_library": the count gkrhip_init put into the
 * process's GPU_MAX_HW_QUEUES (0: it found the variable set -- "hw_queues_from_environment" -- or was told to leave it
 * alone); the runtime reads the variable when it initialises, so the setting only takes effect if the library made the
 * process's first HIP call; "arena_busy_releases": see gkrhip_set_option, "arena_check";
 * "group_launches_wanted" / "group_launches_made": the kernel launches the proofs of gkrhip_mimc_session_prove_group asked for, and
 * the combined launches that served them (wanted / made = proofs per launch); "coalesced_proofs": gkrhip_mimc_session_prove calls that
 * were proven in a group formed from single calls.  Unknown name: error. */
int gkrhip_profile_counter(const char *name, uint64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* GKRHIP_H */
