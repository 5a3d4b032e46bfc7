//go:build gkrhip && (amd64 || arm64)

// GPU bodies of the G1 multi-scalar multiplications of the Groth16 part of the gadget's prover
// (prover/gadget/prove.go:76,91 krsNotGkr / KrsPrivNotGkr over pk.privKNotGkr; :189 bs1 over pk.G1.B; :202 ar over pk.G1.A;
// :221 krs2 over pk.G1.Z with the h of computeH).  Drop into gkr-mimc/prover/gadget/ and replace the MultiExp calls named
// below by these helpers; multiExpG2Jac does the same for the G2 MSM of :277 (Bs over pk.G2.B).  Uncompiled here (no Go toolchain in the build image); the
// entry points are exercised through the C ABI by tests/test_gpu_msm.py.
//
// The bases of these calls are vectors of the proving key -- the same for every proof -- so they are uploaded once
// (g1BasesOf caches a handle per slice) and an MSM moves only its scalars.  Scalars arrive in REGULAR form exactly as the
// reference prepares them (FromMont at prove.go:66,116-120,355-357): no flag needed.
package gadget

import (
	"sync"
	"unsafe"

	"github.com/consensys/gkr-mimc/gkrhip"
	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

var (
	g1BasesMu    sync.Mutex
	g1BasesCache = map[*bn254.G1Affine]*gkrhip.G1Bases{} // keyed by the address of the slice's first point
)

// FixedBaseFromG1 / FixedBaseFromG2: key vectors of at least this many points get fixed-base tables when they are uploaded
// (gkrhip_msm_g1_precompute / _g2_precompute: the window multiples of every point once, 12-13 x the vector's size in HBM; then 12
// or 13 additions per scalar instead of 16).  Measured on one MI355X: G1 pays from 2^18 points (2^22: 6.85 -> 6.19 ms, 2^24:
// 26.7 -> 20.8), G2 from 2^20 (2^22: 19.7 -> 16.3); the back half of ComputeGroth16Proof at 2^22: 75.5 -> 62.6 ms.
// 0 switches the tables off (the vector itself only: 64 / 128 bytes per point).
var (
	FixedBaseFromG1 = 1 << 18
	FixedBaseFromG2 = 1 << 20
)

// g1BasesOf returns the device-resident copy of a proving-key vector, uploading it on first use.
//
// The cache is keyed by the address and length of the slice: it is meant for proving-key vectors, which are written once
// when the key is loaded and never again.  A key that is RELOADED into the same backing array, or mutated in place, must be
// announced with ForgetBases (below) -- the device copy would otherwise be stale and the proofs wrong without any error.
// Entries hold HBM (64 B per point, 128 B for G2) until forgotten.
func g1BasesOf(points []bn254.G1Affine) *gkrhip.G1Bases {
	g1BasesMu.Lock()
	defer g1BasesMu.Unlock()
	key := &points[0]
	if b, ok := g1BasesCache[key]; ok && b.Len() == len(points) {
		return b
	}
	b := gkrhip.NewG1Bases(unsafe.Pointer(&points[0]), len(points))
	if FixedBaseFromG1 > 0 && len(points) >= FixedBaseFromG1 {
		b.Precompute(0)
	}
	g1BasesCache[key] = b
	return b
}

// ForgetBases drops the device copies of the given proving-key vectors (and frees their HBM): call it when a key is unloaded,
// reloaded or changed in place.  The next MSM over such a slice uploads it again.
func ForgetBases(g1 [][]bn254.G1Affine, g2 [][]bn254.G2Affine) {
	g1BasesMu.Lock()
	for _, p := range g1 {
		if len(p) > 0 {
			if b, ok := g1BasesCache[&p[0]]; ok {
				b.Free()
				delete(g1BasesCache, &p[0])
			}
		}
	}
	g1BasesMu.Unlock()
	g2BasesMu.Lock()
	for _, p := range g2 {
		if len(p) > 0 {
			if b, ok := g2BasesCache[&p[0]]; ok {
				b.Free()
				delete(g2BasesCache, &p[0])
			}
		}
	}
	g2BasesMu.Unlock()
}

// sameLength is gnark-crypto's own precondition (MultiExp returns "len(points) != len(scalars)"): the helpers below have no
// error return -- their call sites discard MultiExp's -- so a mismatch panics, as every other misuse of the shims does.
func sameLength(points, scalars int) {
	if points != scalars {
		panic("MultiExp: len(points) != len(scalars)")
	}
}

// multiExpG1Affine replaces `res.MultiExp(points, scalars, ecc.MultiExpConfig{...})` for a bn254.G1Affine receiver
// (prove.go:76,91).
func multiExpG1Affine(res *bn254.G1Affine, points []bn254.G1Affine, scalars []fr.Element) {
	sameLength(len(points), len(scalars))
	if len(scalars) == 0 {
		*res = bn254.G1Affine{}
		return
	}
	g1BasesOf(points).MultiExp(unsafe.Pointer(res), scalars, false)
}

// multiExpG1Jac replaces `res.MultiExp(points, scalars, ecc.MultiExpConfig{...})` for a bn254.G1Jac receiver
// (prove.go:189,202,221): the sum comes back affine and is lifted with Z = 1 -- every later use (AddMixed, AddAssign,
// ScalarMultiplication, FromJacobian: prove.go:194-196,207-210,236-262) is independent of the Jacobian representative.
func multiExpG1Jac(res *bn254.G1Jac, points []bn254.G1Affine, scalars []fr.Element) {
	var aff bn254.G1Affine
	multiExpG1Affine(&aff, points, scalars)
	res.FromAffine(&aff)
}

var (
	g2BasesMu    sync.Mutex
	g2BasesCache = map[*bn254.G2Affine]*gkrhip.G2Bases{}
)

func g2BasesOf(points []bn254.G2Affine) *gkrhip.G2Bases {
	g2BasesMu.Lock()
	defer g2BasesMu.Unlock()
	key := &points[0]
	if b, ok := g2BasesCache[key]; ok && b.Len() == len(points) {
		return b
	}
	b := gkrhip.NewG2Bases(unsafe.Pointer(&points[0]), len(points))
	if FixedBaseFromG2 > 0 && len(points) >= FixedBaseFromG2 {
		b.Precompute(0)
	}
	g2BasesCache[key] = b
	return b
}

// multiExpG2Jac replaces `Bs.MultiExp(pk.G2.B, wireValuesB, ecc.MultiExpConfig{...})` (prove.go:277): the affine sum lifted
// with Z = 1 (the later AddAssign / AddMixed / FromJacobian, prove.go:281-285, do not depend on the representative).
func multiExpG2Jac(res *bn254.G2Jac, points []bn254.G2Affine, scalars []fr.Element) {
	var aff bn254.G2Affine
	sameLength(len(points), len(scalars))
	if len(scalars) > 0 {
		g2BasesOf(points).MultiExp(unsafe.Pointer(&aff), scalars, false)
	}
	res.FromAffine(&aff)
}

// multiExpG1G2Jac replaces the PAIR `bs1.MultiExp(pk.G1.B, wireValuesB, cfg)` (prove.go:189) and
// `Bs.MultiExp(pk.G2.B, wireValuesB, cfg)` (prove.go:277): both are over wireValuesB, so one call uploads, decodes and sorts
// the scalars once and runs both bucket sums on that order.  In ComputeGroth16Proof computeBS1 and computeBS2 then wait on one
// shared sync.Once around this call instead of issuing an MSM each.
func multiExpG1G2Jac(res1 *bn254.G1Jac, res2 *bn254.G2Jac, points1 []bn254.G1Affine, points2 []bn254.G2Affine, scalars []fr.Element) {
	var a1 bn254.G1Affine
	var a2 bn254.G2Affine
	sameLength(len(points1), len(scalars))
	sameLength(len(points2), len(scalars))
	if len(scalars) > 0 {
		gkrhip.MultiExpG1G2(unsafe.Pointer(&a1), unsafe.Pointer(&a2), g1BasesOf(points1), g2BasesOf(points2), scalars, false)
	}
	res1.FromAffine(&a1)
	res2.FromAffine(&a2)
}
