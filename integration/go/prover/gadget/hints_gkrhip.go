//go:build gkrhip

// GPU body of GkrProverHint.Call (prover/gadget/hints.go:197-233): same signature, same input order
// (qPrime || inputs... || outputs, io_store.go:117-136) and same output order (GkrProofToVec, hints.go:236-271);
// only the assign+prove step and the bulk big.Int conversions change.  Drop into gkr-mimc/prover/gadget/ and move the
// pure-Go Call into a file of its own tagged `//go:build !gkrhip` (types, NbOutputs, GkrProofToVec, `debug` stay in hints.go).  Uncompiled here (no Go toolchain in the build image).
package gadget

import (
	"math/big"

	"github.com/consensys/gkr-mimc/common"
	"github.com/consensys/gkr-mimc/gkrhip"
	gkrNative "github.com/consensys/gkr-mimc/gkr"
	"github.com/consensys/gkr-mimc/poly"
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Call computes the GKR proof on the GPU and writes it into oups in GkrProofToVec order
func (h *GkrProverHint) Call(_ ecc.ID, inputsBI []*big.Int, oups []*big.Int) error {
	bN := common.Log2Ceil(h.g.ioStore.Index())
	paddedIndex := 1 << bN

	drain := make([]fr.Element, len(inputsBI))
	for i := range drain {
		drain[i].SetBigInt(inputsBI[i])
	}
	inputs := make([]poly.MultiLin, h.g.Circuit.InputArity())
	qPrime, drain := drain[:bN], drain[bN:]
	for i := range inputs {
		inputs[i], drain = drain[:paddedIndex], drain[paddedIndex:]
	}
	outputs, drain := drain[:paddedIndex], drain[paddedIndex:]
	common.Assert(len(drain) == 0, "The drain was expected to emptied but there remains %v elements", len(drain))

	// Assign + Prove on the device (hints.go:220-222).  The generic path works for any circuit of library gates;
	// examples.MimcCircuit has a one-call form that also overlaps the download of the outputs with the proof.
	t := common.NewTimer("gkr prover hint")
	var flat []fr.Element
	if len(inputs) == 2 && len(h.g.Circuit) == 94 {
		flat = gkrhip.ProveMimc(bN, inputs[0], inputs[1], qPrime, nil)
	} else {
		flat = gkrNative.FlatFromProof(gkrNative.Prove(h.g.Circuit, h.g.Circuit.Assign(inputs...), qPrime))
	}
	t.Close()

	if debug {
		valid := gkrNative.Verify(h.g.Circuit, gkrNative.ProofFromFlat(h.g.Circuit, bN, flat), inputs, outputs, qPrime)
		common.Assert(valid == nil, "GKR proof was wrong - Bug in proof generation - %v", valid)
	}

	// GkrProofToVec: the flat order IS the library's order; one bulk Montgomery -> regular pass on the device
	// instead of len(flat) ToBigIntRegular calls
	if len(flat) != len(oups) {
		panic("expected to have written the entire buffer")
	}
	gkrhip.ToRegular(flat)
	for i := range flat {
		oups[i].SetBits([]big.Word{big.Word(flat[i][0]), big.Word(flat[i][1]), big.Word(flat[i][2]), big.Word(flat[i][3])})
	}
	return nil
}
