//go:build gkrhip && (amd64 || arm64)

// Package gkrhip is the one cgo package of the MI355X back end: Go-typed wrappers over include/gkrhip.h
// (libgkrhip.so).  Every []fr.Element is passed as unsafe.Pointer(&s[0]): gnark-crypto's fr.Element is
// [4]uint64 little-endian Montgomery limbs, canonical, which is exactly the memory image the library reads and
// writes.  Non-zero return codes become panics carrying gkrhip_last_error(), as the reference's prover side
// panics (sumcheck/prover.go:54,114; gkr/prover.go:84).
//
// STATUS: shipped as source.  The image this library is developed in has no Go toolchain, so this file has not
// been compiled; the same entry points, with the same argument meaning, are exercised from compiled C++
// (tests/cpp/test_abi_gkr.cpp) and from Python ctypes (gkr-mimc_amd/prover.py).
//
// It imports nothing of gkr-mimc (poly imports this package; circuit imports poly), so the mapping of a
// circuit.Gate to a library gate lives in the sumcheck shim.
package gkrhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../gkr-mimc_amd -lgkrhip -Wl,-rpath,${SRCDIR}/../../../gkr-mimc_amd
#include <stdlib.h>
#include <string.h>
#include "gkrhip.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Library gate ids (include/gkrhip.h).  Further gates come from RegisterGate.
const (
	GateIdentity = int(C.GKRHIP_GATE_IDENTITY)
	GateCipher   = int(C.GKRHIP_GATE_CIPHER)
	GateAdd      = int(C.GKRHIP_GATE_ADD)
)

// MaxGateInputs is the largest len(Layer.In) the library's gate descriptors express.
const MaxGateInputs = int(C.GKRHIP_MAX_GATE_INPUTS)

// must turns a non-zero return code into a panic carrying the message of THAT failure.  The message is asked for by
// code (gkrhip_last_error_r), not from thread-local state: between the failing cgo call and this one the goroutine may
// have been moved to another OS thread, whose own "last error" would be empty or somebody else's.
func must(rc C.int) {
	if rc != 0 {
		panic("gkrhip: " + errorText(rc))
	}
}

func errorText(rc C.int) string {
	var buf [1024]C.char
	C.gkrhip_last_error_r(rc, &buf[0], C.size_t(len(buf)))
	return C.GoString(&buf[0])
}

// ptr is the address of the first limb of a slice of field elements (nil for an empty slice).
func ptr(s []fr.Element) *C.uint64_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&s[0]))
}

func ptr1(e *fr.Element) *C.uint64_t {
	if e == nil {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(e))
}

// Init selects the GPU (idempotent).  Every other call initialises device 0 on first use.
func Init(device int) { must(C.gkrhip_init(C.int(device))) }

// Fold is (*poly.MultiLin).Fold's body: in place, the folded table is table[:len/2] (poly/multilin.go:19-36).
func Fold(table []fr.Element, r *fr.Element) {
	must(C.gkrhip_fold(ptr(table), C.size_t(len(table)), ptr1(r)))
}

// Evaluate is poly.MultiLin.Evaluate (poly/multilin.go:59-66); the table is left untouched.
func Evaluate(table []fr.Element, coordinates []fr.Element) (res fr.Element) {
	must(C.gkrhip_evaluate(ptr1(&res), ptr(table), C.size_t(len(table)), ptr(coordinates), C.int(len(coordinates))))
	return
}

// EqTable is poly.FoldedEqTable (poly/eq.go:41-59): fills out[:1<<len(qPrime)].
func EqTable(out []fr.Element, qPrime []fr.Element, multiplier *fr.Element) {
	must(C.gkrhip_eq_table(ptr(out), ptr(qPrime), C.int(len(qPrime)), ptr1(multiplier)))
}

// ChunkOfEqTable is poly.ChunkOfEqTable (poly/eq.go:61-89): fills chunk chunkID of table (1<<len(qPrime) elements).
func ChunkOfEqTable(table []fr.Element, chunkID, chunkSize int, qPrime []fr.Element, multiplier *fr.Element) {
	must(C.gkrhip_chunk_of_eq_table(ptr(table), C.size_t(chunkID), C.size_t(chunkSize), ptr(qPrime), C.int(len(qPrime)),
		ptr1(multiplier)))
}

// GateDesc describes a gate of the family the kernels evaluate:
//
//	out = (sum of the inputs selected by SumMask + Ark)^Power,  Power = 1 or 7
//
// (IdentityGate: one input, Power 1; CipherGate: two inputs, Power 7).  Ark is per layer, not part of the gate.
type GateDesc struct {
	ID      string // circuit.Gate.ID()
	NIn     int    // len(Layer.In), 1..MaxGateInputs
	SumMask uint   // bit k set: input k enters the sum
	Power   int    // 1 or 7
}

// RegisterGate adds a gate to the library's registry and returns its id (the same descriptor registered twice
// yields the same id).
func RegisterGate(d GateDesc) int {
	var cd C.gkrhip_gate_desc
	id := C.CString(d.ID)
	defer C.free(unsafe.Pointer(id))
	C.strncpy(&cd.id[0], id, C.size_t(len(cd.id)-1))
	cd.n_in = C.int(d.NIn)
	cd.sum_mask = C.uint(d.SumMask)
	cd.power = C.int(d.Power)
	var out C.int
	must(C.gkrhip_gate_register(&cd, &out))
	return int(out)
}

// GateEvalBatch is circuit.Gate.EvalBatch (circuit/gates.go:16): res[i] = gate(xs[0][i], xs[1][i], ...).
func GateEvalBatch(gate int, ark *fr.Element, res []fr.Element, xs ...[]fr.Element) {
	var pin runtime.Pinner
	defer pin.Unpin()
	cx := make([]*C.uint64_t, len(xs))
	for i := range xs {
		cx[i] = ptr(xs[i])
		pin.Pin(&xs[i][0])
	}
	must(C.gkrhip_gate_eval_batch(C.int(gate), ptr1(ark), ptr(res), (**C.uint64_t)(unsafe.Pointer(&cx[0])), C.int(len(xs)),
		C.size_t(len(res))))
}

// SumcheckProve is sumcheck.Prove's body (sumcheck/prover.go:46-90) on host tables.  X is NOT consumed.
// proof is round-major: proof[k*(degree+2) : (k+1)*(degree+2)] are the coefficients of round k, low to high.
func SumcheckProve(gate int, degree int, ark *fr.Element, X [][]fr.Element, qPrimes [][]fr.Element, claims []fr.Element,
) (proof, challenges, finalClaims []fr.Element) {
	bN := len(qPrimes[0])
	flatQ := make([]fr.Element, 0, len(qPrimes)*bN)
	for _, q := range qPrimes {
		flatQ = append(flatQ, q...)
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	cx := make([]*C.uint64_t, len(X))
	for i := range X {
		cx[i] = ptr(X[i])
		pin.Pin(&X[i][0]) // the table pointers sit in Go memory handed to C: pin what they point to
	}
	nCoeff := degree + 2
	proof = make([]fr.Element, bN*nCoeff+1)
	challenges = make([]fr.Element, bN+1)
	finalClaims = make([]fr.Element, len(X)+1)
	must(C.gkrhip_sumcheck_prove(C.int(gate), ptr1(ark), C.int(len(X)), C.int(bN), (**C.uint64_t)(unsafe.Pointer(&cx[0])),
		ptr(flatQ), C.int(len(qPrimes)), ptr(claims), C.int(len(claims)), ptr(proof), ptr(challenges), ptr(finalClaims)))
	return proof[:bN*nCoeff], challenges[:bN], finalClaims
}

// SumcheckVerify is sumcheck.Verify's body (sumcheck/verifier.go:28-56): proof is round-major, nCoeff coefficients per
// round.  A failed round check comes back as the reference's error ("at round i verifier eval at 0 + 1 = ... ||
// expected = ..."); bad arguments panic like every other misuse.
func SumcheckVerify(claims, proof []fr.Element, bN, nCoeff int) (challenges []fr.Element, finalClaim, recombChal fr.Element, err error) {
	challenges = make([]fr.Element, bN+1)
	if nCoeff < 1 {
		nCoeff = 1
	}
	// the round check's message is the calling thread's last error: keep the goroutine on its thread for the two calls
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := C.gkrhip_sumcheck_verify(ptr(claims), C.int(len(claims)), ptr(proof), C.int(bN), C.int(nCoeff), ptr(challenges),
		ptr1(&finalClaim), ptr1(&recombChal))
	if rc > 0 {
		return nil, fr.Element{}, fr.Element{}, errors.New(C.GoString(C.gkrhip_last_error()))
	}
	must(rc)
	return challenges[:bN], finalClaim, recombChal, nil
}

// ComputeH is computeH's body (prover/gadget/prove.go:308-359): the three inverse FFTs, the three coset FFTs, the pointwise
// (a*b - c) * (-2)^-1 and the inverse coset FFT over fft.NewDomain(cardinality, 1, .) on the device.  a, b, c are the solved
// R1CS vectors (Montgomery elements, len <= cardinality; the library pads with zeros as computeH does); the result has
// `cardinality` elements holding REGULAR-form values at bit-reversed positions, exactly what the reference returns after its
// FromMont loop (prove.go:352-356).
func ComputeH(a, b, c []fr.Element, cardinality uint64) []fr.Element {
	if len(b) != len(a) || len(c) != len(a) {
		panic("gkrhip: computeH: a, b, c differ in length")
	}
	h := make([]fr.Element, cardinality)
	must(C.gkrhip_compute_h(ptr(h), ptr(a), ptr(b), ptr(c), C.size_t(len(a)), C.size_t(cardinality)))
	return h
}

// Layer mirrors circuit.Layer for the library: Gate < 0 marks an input layer.
type Layer struct {
	Gate int
	In   []int
	Ark  fr.Element
}

// Session is a circuit with its assignment resident on the GPU (gkrhip_session).
type Session struct {
	h  *C.gkrhip_session
	bN int
}

func cLayers(layers []Layer) []C.gkrhip_layer {
	cl := make([]C.gkrhip_layer, len(layers))
	for i, l := range layers {
		if len(l.In) > MaxGateInputs {
			panic("gkrhip: a layer has more inputs than the library's gate descriptors express")
		}
		cl[i].gate = C.int(l.Gate)
		cl[i].n_in = C.int(len(l.In))
		for k, v := range l.In {
			cl[i].in[k] = C.int(v)
		}
		for k := 0; k < 4; k++ {
			cl[i].ark[k] = C.uint64_t(l.Ark[k])
		}
	}
	return cl
}

// goLayers converts the library's layer descriptions back to the Go mirror.
func goLayers(cl []C.gkrhip_layer) []Layer {
	out := make([]Layer, len(cl))
	for i := range cl {
		out[i].Gate = int(cl[i].gate)
		out[i].In = make([]int, int(cl[i].n_in))
		for k := range out[i].In {
			out[i].In[k] = int(cl[i].in[k])
		}
		for k := 0; k < 4; k++ {
			out[i].Ark[k] = uint64(cl[i].ark[k])
		}
	}
	return out
}

// GmimcCircuit is the build-defined circuit of one GMiMC compression, out = GMimcT{t}.UpdateInplace(state, block)[0]
// (hash/gmimc.go:52-65), t = 2, 4 or 8.  Input layer k is state[j] when inputMap[k] = j < t, block[j-t] otherwise.
func GmimcCircuit(t int) (layers []Layer, inputMap []int) {
	n := C.gkrhip_gmimc_circuit(C.int(t), nil, 0, nil)
	if n < 0 {
		must(n)
	}
	cl := make([]C.gkrhip_layer, int(n))
	im := make([]C.int, 2*t)
	must(C.gkrhip_gmimc_circuit(C.int(t), &cl[0], n, &im[0]) - n)
	layers = goLayers(cl)
	for _, l := range layers {
		if l.Gate < 0 {
			inputMap = append(inputMap, int(im[len(inputMap)]))
		}
	}
	return layers, inputMap
}

// GmimcHashCircuit is the whole sponge GMimcT{t}.Hash(msg) (hash/gmimc.go:29-49) for messages of nblocks*t elements
// as one circuit.  Input layer k is msg[inputMap[k]].
func GmimcHashCircuit(t, nblocks int) (layers []Layer, inputMap []int) {
	n := C.gkrhip_gmimc_hash_circuit(C.int(t), C.int(nblocks), nil, 0, nil)
	if n < 0 {
		must(n)
	}
	cl := make([]C.gkrhip_layer, int(n))
	im := make([]C.int, t*nblocks)
	must(C.gkrhip_gmimc_hash_circuit(C.int(t), C.int(nblocks), &cl[0], n, &im[0]) - n)
	layers = goLayers(cl)
	for _, l := range layers {
		if l.Gate < 0 {
			inputMap = append(inputMap, int(im[len(inputMap)]))
		}
	}
	return layers, inputMap
}

// NewSession builds the circuit on the device (circuit.BuildCircuit's rules are enforced by the library).
func NewSession(layers []Layer, bN int) *Session {
	cl := cLayers(layers)
	s := &Session{bN: bN}
	must(C.gkrhip_session_create(&s.h, &cl[0], C.int(len(cl)), C.int(bN)))
	runtime.SetFinalizer(s, func(s *Session) { s.Close() })
	return s
}

// LoadInput uploads input layer `index` (2^bN elements).
func (s *Session) LoadInput(index int, table []fr.Element) {
	must(C.gkrhip_session_load_input(s.h, C.int(index), ptr(table)))
}

// Assign is circuit.Circuit.Assign on the device (circuit/assignment.go:12-32).
func (s *Session) Assign() { must(C.gkrhip_mimc_session_assign(s.h)) }

// ProofLen is GkrProverHint.NbOutputs for this circuit and size (prover/gadget/hints.go:76-116).
func (s *Session) ProofLen() int { return int(C.gkrhip_session_proof_len(s.h)) }

// Prove is gkr.Prove (gkr/prover.go:21-47); the result is the flat proof in GkrProofToVec order
// (prover/gadget/hints.go:236-271), still in Montgomery form.  The resident assignment is not consumed.
func (s *Session) Prove(qPrime []fr.Element) []fr.Element {
	flat := make([]fr.Element, s.ProofLen())
	must(C.gkrhip_mimc_session_prove(s.h, ptr(qPrime), ptr(flat)))
	return flat
}

// MaxGroup is the largest number of sessions ProveGroup takes.
const MaxGroup = 8

// ProveGroup is gkr.Prove (gkr/prover.go:21-47) for up to MaxGroup sessions of the same circuit and size in ONE cgo call: proof i
// is, bit for bit, ss[i].Prove(qPrimes[i]); the library proves them in lock-step on the calling thread and sends the round kernels
// of all of them to the GPU as one launch (gkrhip_mimc_session_prove_group).  Where the reference's host proves many small
// statements from a goroutine each, one goroutine per three of them is the faster shape here: bN = 20, 72 in flight, 81 M against
// 64 M hashes/s (profiles/r06_proof_groups.txt); from 2^22 hashes per proof on, plain Prove from a goroutine each is as fast or faster.
func ProveGroup(ss []*Session, qPrimes [][]fr.Element) [][]fr.Element {
	n := len(ss)
	if n == 0 || n > MaxGroup || n != len(qPrimes) {
		panic("gkrhip: ProveGroup takes 1..8 sessions and as many points")
	}
	flats := make([][]fr.Element, n)
	// cgo: no Go pointers to Go pointers -- the three arrays of pointers live in C memory, what they point to is pinned for the call
	word := C.size_t(unsafe.Sizeof(uintptr(0)))
	arr := C.malloc(3 * MaxGroup * word)
	defer C.free(arr)
	hs := (*[MaxGroup]*C.gkrhip_session)(arr)
	qs := (*[MaxGroup]*C.uint64_t)(unsafe.Add(arr, MaxGroup*int(word)))
	fs := (*[MaxGroup]*C.uint64_t)(unsafe.Add(arr, 2*MaxGroup*int(word)))
	var pin runtime.Pinner
	defer pin.Unpin()
	for i, s := range ss {
		flats[i] = make([]fr.Element, s.ProofLen())
		hs[i] = s.h
		qs[i] = ptr(qPrimes[i])
		fs[i] = ptr(flats[i])
		if len(qPrimes[i]) > 0 {
			pin.Pin(&qPrimes[i][0])
		}
		pin.Pin(&flats[i][0])
	}
	must(C.gkrhip_mimc_session_prove_group(C.int(n), &hs[0], &qs[0], &fs[0], nil))
	return flats
}

// Outputs downloads the assignment of the last layer.
func (s *Session) Outputs() []fr.Element {
	out := make([]fr.Element, 1<<s.bN)
	must(C.gkrhip_mimc_session_outputs(s.h, ptr(out)))
	return out
}

// Verify is gkr.Verify against the resident input and output tables (gkr/verifier.go:15-59).
func (s *Session) Verify(qPrime, flat []fr.Element) error {
	rc := C.gkrhip_mimc_session_verify(s.h, ptr(qPrime), ptr(flat))
	if rc > 0 {
		return errors.New(C.GoString(C.gkrhip_last_error()))
	}
	must(rc)
	return nil
}

// Close releases the device tables.
func (s *Session) Close() {
	if s.h != nil {
		C.gkrhip_mimc_session_destroy(s.h)
		s.h = nil
	}
}

// GateDegree is circuit.Gate.Degree() of a library gate (the descriptor's power).
func GateDegree(gate int) int {
	var d C.gkrhip_gate_desc
	must(C.gkrhip_gate_lookup(C.int(gate), &d))
	return int(d.power)
}

// ProofLen is GkrProverHint.NbOutputs (prover/gadget/hints.go:76-116) for a circuit given as a layer list.
func ProofLen(layers []Layer, bN int) int {
	outs := make([]int, len(layers))
	n := bN // the output layer's qPrime
	for _, l := range layers {
		for _, p := range l.In {
			outs[p]++
		}
		if l.Gate >= 0 {
			n += bN * (GateDegree(l.Gate) + 2)
		}
	}
	for _, o := range outs {
		n += o + bN*o
	}
	return n
}

// Prove is Circuit.Assign + gkr.Prove for any circuit of library gates on host tables in one call; the flat proof is
// in GkrProofToVec order.  outputs may be nil.
func Prove(layers []Layer, bN int, inputs [][]fr.Element, qPrime, outputs []fr.Element) []fr.Element {
	cl := cLayers(layers)
	var pin runtime.Pinner
	defer pin.Unpin()
	ci := make([]*C.uint64_t, len(inputs))
	for i := range inputs {
		ci[i] = ptr(inputs[i])
		pin.Pin(&inputs[i][0])
	}
	flat := make([]fr.Element, ProofLen(layers, bN))
	must(C.gkrhip_gkr_prove(&cl[0], C.int(len(cl)), C.int(bN), (**C.uint64_t)(unsafe.Pointer(&ci[0])), C.int(len(ci)),
		ptr(qPrime), ptr(flat), ptr(outputs)))
	return flat
}

// Verify is gkr.Verify (gkr/verifier.go:15-59) for any circuit of library gates on host tables: the sumcheck
// verifiers and the claim bookkeeping run on the host, MultiLin.Evaluate of inputs and outputs on the device.
func Verify(layers []Layer, bN int, flat []fr.Element, inputs [][]fr.Element, outputs, qPrime []fr.Element) error {
	cl := cLayers(layers)
	var pin runtime.Pinner
	defer pin.Unpin()
	ci := make([]*C.uint64_t, len(inputs))
	for i := range inputs {
		ci[i] = ptr(inputs[i])
		pin.Pin(&inputs[i][0])
	}
	rc := C.gkrhip_gkr_verify(&cl[0], C.int(len(cl)), C.int(bN), ptr(flat), (**C.uint64_t)(unsafe.Pointer(&ci[0])),
		C.int(len(ci)), ptr(outputs), ptr(qPrime))
	if rc > 0 {
		return errors.New(C.GoString(C.gkrhip_last_error()))
	}
	must(rc)
	return nil
}

// ProveMimc is Circuit.Assign + gkr.Prove for examples.MimcCircuit in one call, what GkrProverHint.Call times
// (prover/gadget/hints.go:220-222).  outputs may be nil.
func ProveMimc(bN int, in0, in1, qPrime, outputs []fr.Element) []fr.Element {
	flat := make([]fr.Element, int(C.gkrhip_mimc_proof_len(C.int(bN))))
	must(C.gkrhip_gkr_prove_mimc(C.int(bN), ptr(in0), ptr(in1), ptr(qPrime), ptr(flat), ptr(outputs)))
	return flat
}

// ProveMimcRegular is ProveMimc with every buffer in REGULAR form ([4]uint64 little-endian words of the value itself, what
// big.Int.Bits() holds): the hint's big.Int conversions become word copies, the Montgomery conversions ride on the library's
// boundary transposition.  in0, in1, qPrime and the results are NOT valid fr.Elements in the Montgomery sense -- they are
// carried in []fr.Element only for its memory layout.
func ProveMimcRegular(bN int, in0, in1, qPrime, outputs []fr.Element) []fr.Element {
	flat := make([]fr.Element, int(C.gkrhip_mimc_proof_len(C.int(bN))))
	must(C.gkrhip_gkr_prove_mimc_regular(C.int(bN), ptr(in0), ptr(in1), ptr(qPrime), ptr(flat), ptr(outputs)))
	return flat
}

// VerifyMimc is gkr.Verify for examples.MimcCircuit on host tables.
func VerifyMimc(bN int, flat, in0, in1, outputs, qPrime []fr.Element) error {
	rc := C.gkrhip_gkr_verify_mimc(C.int(bN), ptr(flat), ptr(in0), ptr(in1), ptr(outputs), ptr(qPrime))
	if rc > 0 {
		return errors.New(C.GoString(C.gkrhip_last_error()))
	}
	must(rc)
	return nil
}

// ToRegular / FromRegular convert a slice in place between Montgomery limbs and the regular value as four
// little-endian words (what big.Int.SetBits / Bits exchange on amd64): the bulk form of ToBigIntRegular / SetBigInt
// (prover/gadget/hints.go:202-205,236-271).
func ToRegular(s []fr.Element)   { must(C.gkrhip_to_regular(ptr(s), C.size_t(len(s)))) }
func FromRegular(s []fr.Element) { must(C.gkrhip_from_regular(ptr(s), C.size_t(len(s)))) }

// MimcPermutationBatch: out[i] = hash.MimcKeyedPermutation(x[i], key[i]) (hash/mimc.go:31-39), the body of
// HashHint.Call for a whole batch (prover/gadget/hints.go:134-145).
func MimcPermutationBatch(out, x, key []fr.Element) {
	must(C.gkrhip_mimc_permutation_batch(ptr(out), ptr(x), ptr(key), C.size_t(len(x))))
}

// ---- G1 multi-scalar multiplication (include/gkrhip.h: gkrhip_msm_g1 and friends) ---------------------------------
// A bn254.G1Affine is {X, Y fp.Element} = 8 uint64 in Montgomery form, infinity = (0, 0): the memory image the library
// reads; scalars are fr.Elements in REGULAR form unless scalarsMont is set (ecc.MultiExpConfig.ScalarsMont).  This file
// does not import gnark-crypto's bn254 package (it would only be needed for the type): the callers pass
// unsafe.Pointer(&points[0]).

// G1Bases is a proving-key vector (pk.G1.A, pk.G1.B, pk.G1.Z, pk.privKNotGkr) resident in HBM.
type G1Bases struct{ h *C.gkrhip_g1_bases }

// NewG1Bases uploads n points once; they stay on the device until Free.
func NewG1Bases(points unsafe.Pointer, n int) *G1Bases {
	b := &G1Bases{}
	must(C.gkrhip_g1_bases_create(&b.h, (*C.uint64_t)(points), C.size_t(n)))
	runtime.SetFinalizer(b, func(b *G1Bases) { b.Free() })
	return b
}

func (b *G1Bases) Free() {
	if b.h != nil {
		C.gkrhip_g1_bases_destroy(b.h)
		b.h = nil
	}
}

func (b *G1Bases) Len() int { return int(C.gkrhip_g1_bases_len(b.h)) }

// Precompute builds the fixed-base tables [2^(c j)] P_i of the handle once (a proving-key vector is the same for every proof):
// MultiExp on the handle then sorts all windows into one bucket space -- 12 or 13 additions per scalar instead of 16.
// c = 0: chosen from the number of points; -1 drops the tables.
func (b *G1Bases) Precompute(c int) { must(C.gkrhip_msm_g1_precompute(b.h, C.int(c))) }

// MultiExp writes sum_i [scalars[i]] bases[i] into out (a *bn254.G1Affine) for the first len(scalars) bases.
func (b *G1Bases) MultiExp(out unsafe.Pointer, scalars []fr.Element, scalarsMont bool) {
	flags := C.int(0)
	if scalarsMont {
		flags = C.GKRHIP_MSM_SCALARS_MONT
	}
	must(C.gkrhip_msm_g1((*C.uint64_t)(out), b.h, ptr(scalars), C.size_t(len(scalars)), flags))
}

// MultiExpG1 is (*G1Affine).MultiExp(points, scalars, config) in one call on host slices (bases uploaded per call).
func MultiExpG1(out, points unsafe.Pointer, scalars []fr.Element, scalarsMont bool) {
	flags := C.int(0)
	if scalarsMont {
		flags = C.GKRHIP_MSM_SCALARS_MONT
	}
	must(C.gkrhip_msm_g1_once((*C.uint64_t)(out), (*C.uint64_t)(points), ptr(scalars), C.size_t(len(scalars)), flags))
}

// BatchScalarMultiplicationG1 is bn254.BatchScalarMultiplicationG1(base, scalars) (prove.go:177): out is a
// []bn254.G1Affine of len(scalars) elements.
func BatchScalarMultiplicationG1(out, base unsafe.Pointer, scalars []fr.Element) {
	must(C.gkrhip_g1_batch_scalar_mul((*C.uint64_t)(out), (*C.uint64_t)(base), ptr(scalars), C.size_t(len(scalars)), 0))
}

// ---- G2: (*G2Jac).MultiExp (prove.go:277) ---------------------------------------------------------------------------
// A bn254.G2Affine is {X, Y fptower.E2{A0, A1 fp.Element}} = 16 uint64 in Montgomery form, infinity = zeros.

// G2Bases is a proving-key vector on G2 (pk.G2.B) resident in HBM.
type G2Bases struct{ h *C.gkrhip_g2_bases }

func NewG2Bases(points unsafe.Pointer, n int) *G2Bases {
	b := &G2Bases{}
	must(C.gkrhip_g2_bases_create(&b.h, (*C.uint64_t)(points), C.size_t(n)))
	runtime.SetFinalizer(b, func(b *G2Bases) { b.Free() })
	return b
}

func (b *G2Bases) Free() {
	if b.h != nil {
		C.gkrhip_g2_bases_destroy(b.h)
		b.h = nil
	}
}

func (b *G2Bases) Len() int { return int(C.gkrhip_g2_bases_len(b.h)) }

// Precompute: as (*G1Bases).Precompute.
func (b *G2Bases) Precompute(c int) { must(C.gkrhip_msm_g2_precompute(b.h, C.int(c))) }

// MultiExp writes sum_i [scalars[i]] bases[i] into out (a *bn254.G2Affine).
func (b *G2Bases) MultiExp(out unsafe.Pointer, scalars []fr.Element, scalarsMont bool) {
	flags := C.int(0)
	if scalarsMont {
		flags = C.GKRHIP_MSM_SCALARS_MONT
	}
	must(C.gkrhip_msm_g2((*C.uint64_t)(out), b.h, ptr(scalars), C.size_t(len(scalars)), flags))
}

// MultiExpShared runs len(g1) G1 sums and len(g2) G2 sums over ONE scalar vector (gkrhip_msm_shared): one upload, one sort.
// outG1[i] is a *bn254.G1Affine, outG2[i] a *bn254.G2Affine.  Every handle holds the same number of points; vectors that the
// proving key stores filtered (pk.G1.A by pk.InfinityA, pk.G1.B / pk.G2.B by pk.InfinityB) are uploaded expanded, with the
// point at infinity at the dropped positions, so that all of them line up with the unfiltered wireValues.
func MultiExpShared(outG1 []unsafe.Pointer, g1 []*G1Bases, outG2 []unsafe.Pointer, g2 []*G2Bases, scalars []fr.Element) {
	h1 := make([]*C.gkrhip_g1_bases, len(g1))
	h2 := make([]*C.gkrhip_g2_bases, len(g2))
	o1 := make([]C.uint64_t, 8*len(g1))
	o2 := make([]C.uint64_t, 16*len(g2))
	for i, b := range g1 {
		h1[i] = b.h
	}
	for i, b := range g2 {
		h2[i] = b.h
	}
	var p1 **C.gkrhip_g1_bases
	var p2 **C.gkrhip_g2_bases
	var q1, q2 *C.uint64_t
	if len(g1) > 0 {
		p1, q1 = &h1[0], &o1[0]
	}
	if len(g2) > 0 {
		p2, q2 = &h2[0], &o2[0]
	}
	must(C.gkrhip_msm_shared(q1, q2, p1, C.size_t(len(g1)), p2, C.size_t(len(g2)), ptr(scalars), C.size_t(len(scalars)), 0))
	for i := range g1 {
		copy(unsafe.Slice((*C.uint64_t)(outG1[i]), 8), o1[8*i:8*i+8])
	}
	for i := range g2 {
		copy(unsafe.Slice((*C.uint64_t)(outG2[i]), 16), o2[16*i:16*i+16])
	}
}

// ReserveLanes creates n lanes ahead of the first burst of concurrent calls (the goroutines of ComputeGroth16Proof lease one
// each; a lane takes ~3 ms to create).  Optional.
func ReserveLanes(n int) { must(C.gkrhip_reserve_lanes(C.int(n))) }

// SetOption: gkrhip_set_option.  Of interest to a host: "layer_check" (default 1: every sumcheck is held against the
// verifier's identities before it is returned and re-run in safe mode if it does not close) and "verify_after_prove"
// (default 0: the one-shot calls run gkr.Verify on their proof first, as the reference's hint does in debug builds,
// prover/gadget/hints.go:224-228).
func SetOption(key string, value int) {
	k := C.CString(key)
	defer C.free(unsafe.Pointer(k))
	must(C.gkrhip_set_option(k, C.long(value)))
}

// Counter: gkrhip_profile_counter, e.g. "layer_check_failures" (sumchecks that did not close and were run again: anything
// but 0 means the device side slipped), "chal_retries", "ahead_round0", "hw_queues_set_by_library".
func Counter(name string) uint64 {
	k := C.CString(name)
	defer C.free(unsafe.Pointer(k))
	var v C.uint64_t
	must(C.gkrhip_profile_counter(k, &v))
	return uint64(v)
}

// PinnedElements returns a []fr.Element of length n in page-locked host memory (gkrhip_host_alloc): uploads from it are plain
// DMA transfers instead of staged copies of pageable memory.  Meant for the vectors handed over on every proof (wireValues and
// its filtered copies, the a / b / c of computeH).  The memory is not known to Go's collector: release it with FreePinned.
func PinnedElements(n int) []fr.Element {
	var p unsafe.Pointer
	must(C.gkrhip_host_alloc(&p, C.size_t(n)*C.size_t(unsafe.Sizeof(fr.Element{}))))
	return unsafe.Slice((*fr.Element)(p), n)
}

// FreePinned releases a slice obtained from PinnedElements.
func FreePinned(s []fr.Element) {
	if len(s) > 0 {
		C.gkrhip_host_free(unsafe.Pointer(&s[0]))
	}
}

// MultiExpG1G2 is `bs1.MultiExp(pk.G1.B, wireValuesB, cfg)` and `Bs.MultiExp(pk.G2.B, wireValuesB, cfg)` (prover/gadget/prove.go:189,277)
// in one call: both sums are over the same scalars, which are uploaded, decoded and sorted once.  outG1 is a *bn254.G1Affine,
// outG2 a *bn254.G2Affine; the two handles hold the same number of points (both vectors are filtered by pk.InfinityB).
func MultiExpG1G2(outG1, outG2 unsafe.Pointer, b1 *G1Bases, b2 *G2Bases, scalars []fr.Element, scalarsMont bool) {
	flags := C.int(0)
	if scalarsMont {
		flags = C.GKRHIP_MSM_SCALARS_MONT
	}
	must(C.gkrhip_msm_g1_g2((*C.uint64_t)(outG1), (*C.uint64_t)(outG2), b1.h, b2.h, ptr(scalars), C.size_t(len(scalars)), flags))
}

// ComputeHMultiExp is `h := computeH(a, b, c, domain)` followed by `krs2.MultiExp(pk.G1.Z, h, cfg)` (prover/gadget/prove.go:128,221)
// in one call with H never leaving the device.  out is a *bn254.G1Affine; h (optional, len = cardinality) also receives H.
func (b *G1Bases) ComputeHMultiExp(out unsafe.Pointer, a, bb, c []fr.Element, cardinality uint64, h []fr.Element) {
	must(C.gkrhip_compute_h_msm_g1((*C.uint64_t)(out), b.h, ptr(a), ptr(bb), ptr(c), C.size_t(len(a)), C.size_t(cardinality), ptr(h)))
}
