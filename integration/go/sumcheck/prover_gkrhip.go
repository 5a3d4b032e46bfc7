//go:build gkrhip && (amd64 || arm64)

// GPU body of sumcheck.Prove (sumcheck/prover.go:46-90).  Drop into gkr-mimc/sumcheck/ and move prover.go's Prove into
// a file of its own tagged `//go:build !gkrhip` (build tags are per file; the helpers of prover.go stay).  Uncompiled here (no Go toolchain in the build image).
package sumcheck

import (
	"fmt"

	"github.com/consensys/gkr-mimc/circuit"
	"github.com/consensys/gkr-mimc/circuit/gates"
	"github.com/consensys/gkr-mimc/gkrhip"
	"github.com/consensys/gkr-mimc/poly"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// LibraryGate maps a circuit.Gate to the library's gate id and the layer's Ark.  CipherGate and IdentityGate are
// built in; any other gate of the family out = (sum of selected inputs + Ark)^{1|7} is registered once with
// gkrhip.RegisterGate and added to the switch (or implements LibraryGater).  Anything else panics: the library has
// no CPU fallback.
func LibraryGate(g circuit.Gate, nIn int) (id int, ark *fr.Element) {
	switch t := g.(type) {
	case *gates.CipherGate:
		return gkrhip.GateCipher, &t.Ark
	case gates.IdentityGate:
		if nIn == 1 {
			return gkrhip.GateIdentity, nil
		}
		// IdentityGate over several tables returns xs[0] (circuit/gates/copy.go:15-22; InitializeMultiInstance
		// passes [L, R]): a registered gate whose sum selects input 0 only
		return gkrhip.RegisterGate(gkrhip.GateDesc{ID: g.ID(), NIn: nIn, SumMask: 1, Power: 1}), nil
	case LibraryGater:
		return t.LibraryGate(nIn)
	}
	panic("gate not supported by libgkrhip: " + g.ID())
}

// LibraryGater is implemented by gates defined outside this repository that the library can evaluate.
type LibraryGater interface {
	LibraryGate(nIn int) (id int, ark *fr.Element)
}

// Prove contains the coordination logic for all workers contributing to the sumcheck proof
func Prove(X []poly.MultiLin, qPrimes [][]fr.Element, claims []fr.Element, gate circuit.Gate) (proof Proof, challenges, finalClaims []fr.Element) {
	bN := len(qPrimes[0])
	for i, x := range X { // same sanity check and panic text as sumcheck/prover.go:52-56
		if len(x) != 1<<bN {
			panic(fmt.Sprintf("inconsistent sizes : bn is %v but table %v has size %v", bN, i, len(x)))
		}
	}
	tables := make([][]fr.Element, len(X))
	for i := range X {
		tables[i] = X[i]
	}
	id, ark := LibraryGate(gate, len(X))
	nCoeff := gate.Degree() + 2
	flat, challenges, finalClaims := gkrhip.SumcheckProve(id, gate.Degree(), ark, tables, qPrimes, claims)
	proof = make(Proof, bN)
	for k := range proof {
		proof[k] = flat[k*nCoeff : (k+1)*nCoeff]
	}
	// The reference consumes X (folds in place and returns the buffers to its pool, prover.go:83-86); the library
	// leaves X untouched.  Callers may not rely on X after Prove either way; give the buffers back as it does.
	for _, x := range X {
		poly.DumpLarge(x)
	}
	return proof, challenges, finalClaims
}

// Verify is sumcheck.Verify (sumcheck/verifier.go:28-56) through the library, so that the package is replaced
// symmetrically (move the pure-Go Verify into a file tagged `//go:build !gkrhip`; recombineMultiClaims stays).  It is
// scalar work -- the Fiat-Shamir chain -- and runs on the host inside libgkrhip; same outputs, same error text.
func Verify(claims []fr.Element, proof Proof) (challenges []fr.Element, finalClaim, recombChal fr.Element, err error) {
	bn := len(proof)
	nCoeff := 0
	if bn > 0 {
		nCoeff = len(proof[0])
	}
	flat := make([]fr.Element, 0, bn*nCoeff)
	for _, p := range proof {
		flat = append(flat, p...)
	}
	challenges, finalClaim, recombChal, err = gkrhip.SumcheckVerify(claims, flat, bn, nCoeff)
	if err != nil {
		return nil, fr.Element{}, fr.Element{}, err
	}
	return challenges, finalClaim, recombChal, nil
}
