//go:build gkrhip && (amd64 || arm64)

// GPU bodies of gkr.Prove (gkr/prover.go:21-47) and gkr.Verify (gkr/verifier.go:15-59).  Drop into gkr-mimc/gkr/
// with `//go:build !gkrhip` on the pure-Go Prove / Verify.  Uncompiled here (no Go toolchain in the build image);
// the flat <-> Proof conversion below is mirrored, and checked against the oracle's per-layer arrays, by
// tests/cpp/test_abi_gkr.cpp (proof_from_flat).
package gkr

import (
	"github.com/consensys/gkr-mimc/circuit"
	"github.com/consensys/gkr-mimc/gkrhip"
	"github.com/consensys/gkr-mimc/poly"
	"github.com/consensys/gkr-mimc/sumcheck"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Proof contains all the data for a GKR to be verified
type Proof struct {
	SumcheckProofs []sumcheck.Proof
	Claims         [][]fr.Element
	QPrimes        [][][]fr.Element
}

// libraryLayers maps circuit.Circuit to the library's layer list (In as given, Out is recomputed by the library
// exactly as BuildCircuit does).  Panics on a gate the library cannot evaluate.
func libraryLayers(c circuit.Circuit) []gkrhip.Layer {
	layers := make([]gkrhip.Layer, len(c))
	for l := range c {
		layers[l].Gate = -1
		if c[l].Gate == nil {
			continue
		}
		id, ark := sumcheck.LibraryGate(c[l].Gate, len(c[l].In))
		layers[l].Gate = id
		layers[l].In = c[l].In
		if ark != nil {
			layers[l].Ark = *ark
		}
	}
	return layers
}

// session uploads the input layers of `a` and recomputes the assignment on the device (Circuit.Assign is a
// deterministic function of the input layers, circuit/assignment.go:12-32, so the resident tables equal `a`).
func session(c circuit.Circuit, inputs []poly.MultiLin, bN int) *gkrhip.Session {
	s := gkrhip.NewSession(libraryLayers(c), bN)
	for l := range inputs {
		s.LoadInput(l, inputs[l])
	}
	s.Assign()
	return s
}

// Prove returns a new prover
func Prove(c circuit.Circuit, a circuit.Assignment, qPrime []fr.Element) (proof Proof) {
	bN := len(qPrime)
	s := session(c, a[:c.InputArity()], bN)
	defer s.Close()
	proof = ProofFromFlat(c, bN, s.Prove(qPrime))
	a.Dump() // the reference's Prove consumes the assignment (InputsOfLayer hands a[pos] to sumcheck.Prove, which dumps it)
	return proof
}

// ProofFromFlat is the inverse of GkrProofToVec (prover/gadget/hints.go:236-271) for elements kept in Montgomery
// form: all sumcheck coefficients layer by layer and round by round, then every layer's claims, then every
// layer's evaluation points.  Layers that never received a claim keep nil slices, as gkr.Prove leaves them.
func ProofFromFlat(c circuit.Circuit, bN int, flat []fr.Element) (proof Proof) {
	n := len(c)
	proof.SumcheckProofs = make([]sumcheck.Proof, n)
	proof.Claims = make([][]fr.Element, n)
	proof.QPrimes = make([][][]fr.Element, n)
	cur := 0
	for l := range c {
		if c[l].Gate == nil {
			continue
		}
		nCoeff := c[l].Gate.Degree() + 2
		proof.SumcheckProofs[l] = make(sumcheck.Proof, bN)
		for k := 0; k < bN; k++ {
			proof.SumcheckProofs[l][k] = flat[cur : cur+nCoeff]
			cur += nCoeff
		}
	}
	for l := range c {
		if len(c[l].Out) > 0 {
			proof.Claims[l] = flat[cur : cur+len(c[l].Out)]
			cur += len(c[l].Out)
		}
	}
	for l := range c {
		slots := len(c[l].Out)
		if l == n-1 {
			slots = 1 // the output layer holds the initial qPrime (gkr/prover.go:31)
		}
		if slots == 0 {
			continue
		}
		proof.QPrimes[l] = make([][]fr.Element, slots)
		for w := 0; w < slots; w++ {
			proof.QPrimes[l][w] = flat[cur : cur+bN]
			cur += bN
		}
	}
	if cur != len(flat) {
		panic("gkrhip: flat proof length does not match the circuit")
	}
	return proof
}

// FlatFromProof is GkrProofToVec without the big.Int conversion.
func FlatFromProof(proof Proof) (flat []fr.Element) {
	for _, layer := range proof.SumcheckProofs {
		for _, round := range layer {
			flat = append(flat, round...)
		}
	}
	for _, layer := range proof.Claims {
		flat = append(flat, layer...)
	}
	for _, layer := range proof.QPrimes {
		for _, qs := range layer {
			flat = append(flat, qs...)
		}
	}
	return flat
}

// Verify runs the whole of gkr.Verify natively: the sumcheck verifiers and the claim bookkeeping on the host, the
// O(N) MultiLin.Evaluate of inputs and outputs on the device (gkr/verifier.go:15-59).
func Verify(c circuit.Circuit, proof Proof, inputs []poly.MultiLin, outputs poly.MultiLin, qPrime []fr.Element) error {
	tables := make([][]fr.Element, len(inputs))
	for i := range inputs {
		tables[i] = inputs[i]
	}
	return gkrhip.Verify(libraryLayers(c), len(qPrime), FlatFromProof(proof), tables, outputs, qPrime)
}
