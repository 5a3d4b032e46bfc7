//go:build gkrhip && (amd64 || arm64)

// GPU body of computeH (prover/gadget/prove.go:308-359), the H part of Groth16's Krs.  Drop into gkr-mimc/prover/gadget/ and
// move the pure-Go computeH into a file of its own tagged `//go:build !gkrhip` (Prove and its goroutines stay where they are:
// they call computeH by name).  Uncompiled here (no Go toolchain in the build image); the entry point is exercised through
// the C ABI by tests/test_gpu_compute_h.py.  The G1 MSMs of Prove (prove.go:76,91,189,202,221) and the G2 MSM (prove.go:277) are in
// msm_gkrhip.go.
package gadget

import (
	"github.com/consensys/gkr-mimc/gkrhip"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/fft"
)

// computeH returns the coefficients of H = (A*B - C) / Z evaluated as the reference does (ifft, coset fft, pointwise
// (a*b - c) * (-2)^-1, coset ifft, FromMont).  The domain's twiddles live on the device (cached per cardinality); only its
// cardinality is read here -- the library derives Generator and FinerGenerator as fft.NewDomain(cardinality, 1, .) does.
func computeH(a, b, c []fr.Element, domain *fft.Domain) []fr.Element {
	return gkrhip.ComputeH(a, b, c, domain.Cardinality)
}
