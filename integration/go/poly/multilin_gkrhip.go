//go:build gkrhip && (amd64 || arm64)

// GPU bodies of the poly functions on the GKR hot path.  Drop this file into gkr-mimc/poly/ and put
// `//go:build !gkrhip` on the pure-Go definitions it replaces (poly/multilin.go:19-23,59-66, poly/eq.go:41-59);
// signatures, results and panics are unchanged.  Uncompiled here (no Go toolchain in the build image).
package poly

import (
	"github.com/consensys/gkr-mimc/gkrhip" // integration/go/gkrhip, vendored next to poly/
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Fold folds the table on its first coordinate using the given value r (poly/multilin.go:19-23)
func (bkt *MultiLin) Fold(r fr.Element) {
	mid := len(*bkt) / 2
	gkrhip.Fold(*bkt, &r)
	*bkt = (*bkt)[:mid]
}

// Evaluate the multilinear polynomial at the given coordinates (poly/multilin.go:59-66)
func (bkt MultiLin) Evaluate(coordinates []fr.Element) fr.Element {
	return gkrhip.Evaluate(bkt, coordinates)
}

// FoldedEqTable ought to start life as a sparse bookkeepingtable (poly/eq.go:41-59)
func FoldedEqTable(preallocated MultiLin, qPrime []fr.Element, multiplier ...fr.Element) MultiLin {
	var m *fr.Element
	if len(multiplier) > 0 {
		m = &multiplier[0]
	}
	gkrhip.EqTable(preallocated, qPrime, m)
	return preallocated
}

// ChunkOfEqTable computes only a chunk of the eqTable for a given chunkSize and chunkID (poly/eq.go:61-89)
func ChunkOfEqTable(preallocatedEq []fr.Element, chunkID, chunkSize int, qPrime []fr.Element, multiplier ...fr.Element) {
	var m *fr.Element
	if len(multiplier) > 0 {
		m = &multiplier[0]
	}
	gkrhip.ChunkOfEqTable(preallocatedEq, chunkID, chunkSize, qPrime, m)
}
