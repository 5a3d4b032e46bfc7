//go:build gkrhip && (amd64 || arm64)

// GPU body of GkrProverHint.Call (prover/gadget/hints.go:197-233): same signature, same input order
// (qPrime || inputs... || outputs, io_store.go:117-136) and same output order (GkrProofToVec, hints.go:236-271);
// only the assign+prove step and the bulk big.Int conversions change.  Drop into gkr-mimc/prover/gadget/ and move the
// pure-Go Call into a file of its own tagged `//go:build !gkrhip` (types, NbOutputs, GkrProofToVec, `debug` stay in hints.go).  Uncompiled here (no Go toolchain in the build image).
package gadget

import (
	"math/big"

	"github.com/consensys/gkr-mimc/circuit"
	"github.com/consensys/gkr-mimc/common"
	"github.com/consensys/gkr-mimc/examples"
	"github.com/consensys/gkr-mimc/gkrhip"
	gkrNative "github.com/consensys/gkr-mimc/gkr"
	"github.com/consensys/gkr-mimc/poly"
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// isMimcCircuit reports whether c is examples.MimcCircuit() (examples/mimc.go:10-37) layer for layer: the same wiring and
// the same gates (Gate.ID() of a CipherGate carries its Ark, circuit/gates/cipher.go:22).  Only then may the one-call
// MiMC entry point stand in for Assign + Prove; a different circuit that merely has 94 layers and two inputs takes the
// generic path.
func isMimcCircuit(c circuit.Circuit) bool {
	ref := examples.MimcCircuit()
	if len(c) != len(ref) {
		return false
	}
	for l := range c {
		if (c[l].Gate == nil) != (ref[l].Gate == nil) || len(c[l].In) != len(ref[l].In) {
			return false
		}
		if c[l].Gate != nil && c[l].Gate.ID() != ref[l].Gate.ID() {
			return false
		}
		for k := range c[l].In {
			if c[l].In[k] != ref[l].In[k] {
				return false
			}
		}
	}
	return true
}

// Call computes the GKR proof on the GPU and writes it into oups in GkrProofToVec order
func (h *GkrProverHint) Call(_ ecc.ID, inputsBI []*big.Int, oups []*big.Int) error {
	bN := common.Log2Ceil(h.g.ioStore.Index())
	paddedIndex := 1 << bN

	// big.Int -> words, no arithmetic: the values stay in REGULAR form and the library converts them to Montgomery form on
	// the device while it transposes the tables (gkrhip_gkr_prove_mimc_regular); the pure-Go path pays one SetBigInt
	// (a Montgomery multiplication) per element here, 2^(bN+1) of them
	drain := make([]fr.Element, len(inputsBI))
	for i := range drain {
		b := inputsBI[i]
		if b.Sign() < 0 || b.Cmp(fr.Modulus()) >= 0 {
			b = new(big.Int).Mod(b, fr.Modulus())
		}
		for k, w := range b.Bits() { // little-endian words; big.Word is 64 bits wide on amd64 / arm64 (the build constraint above)
			drain[i][k] = uint64(w)
		}
	}
	inputs := make([]poly.MultiLin, h.g.Circuit.InputArity())
	qPrime, drain := drain[:bN], drain[bN:]
	for i := range inputs {
		inputs[i], drain = drain[:paddedIndex], drain[paddedIndex:]
	}
	outputs, drain := drain[:paddedIndex], drain[paddedIndex:]
	common.Assert(len(drain) == 0, "The drain was expected to emptied but there remains %v elements", len(drain))

	// Assign + Prove on the device (hints.go:220-222).  examples.MimcCircuit has a one-call form on regular-form buffers;
	// any other circuit of library gates goes through the generic path on Montgomery elements.
	t := common.NewTimer("gkr prover hint")
	var flat []fr.Element // regular form from here on
	if isMimcCircuit(h.g.Circuit) {
		flat = gkrhip.ProveMimcRegular(bN, inputs[0], inputs[1], qPrime, nil)
		if debug { // the verifier takes Montgomery elements: convert copies
			m := func(s []fr.Element) []fr.Element { c := append([]fr.Element{}, s...); gkrhip.FromRegular(c); return c }
			valid := gkrhip.VerifyMimc(bN, m(flat), m(inputs[0]), m(inputs[1]), m(outputs), m(qPrime))
			common.Assert(valid == nil, "GKR proof was wrong - Bug in proof generation - %v", valid)
		}
	} else {
		for i := range inputs {
			gkrhip.FromRegular(inputs[i])
		}
		gkrhip.FromRegular(qPrime)
		flat = gkrNative.FlatFromProof(gkrNative.Prove(h.g.Circuit, h.g.Circuit.Assign(inputs...), qPrime))
		if debug {
			gkrhip.FromRegular(outputs)
			valid := gkrNative.Verify(h.g.Circuit, gkrNative.ProofFromFlat(h.g.Circuit, bN, flat), inputs, outputs, qPrime)
			common.Assert(valid == nil, "GKR proof was wrong - Bug in proof generation - %v", valid)
		}
		gkrhip.ToRegular(flat)
	}
	t.Close()

	// GkrProofToVec: the flat order IS the library's order, and the elements are regular-form words already
	if len(flat) != len(oups) {
		panic("expected to have written the entire buffer")
	}
	for i := range flat {
		oups[i].SetBits([]big.Word{big.Word(flat[i][0]), big.Word(flat[i][1]), big.Word(flat[i][2]), big.Word(flat[i][3])})
	}
	return nil
}
