"""CPU restatement (Python big ints) of computeH -- the H part of Groth16's Krs -- as the reference's prover gadget runs
it (prover/gadget/prove.go:308-359), for SURVEY section 8 row f4.  TEST INFRASTRUCTURE ONLY: nothing in the product
imports this file.

PARITY UNPINNED.  The arithmetic computeH calls lives in a dependency that is NOT under /root/reference:
`github.com/consensys/gnark-crypto v0.6.1-0.20220110145513-493bb1c180d9` (go.mod:7), package `ecc/bn254/fr/fft`
(`Domain`, `NewDomain`, `FFT`, `FFTInverse`, `DIF`/`DIT`); the domain itself is built by the un-vendored gnark fork
(`pkg/gnark`, empty here) as `fft.NewDomain(nbConstraints, 1, true)`.  The reference holds no vector for computeH (no test
calls it with stored outputs), so this restatement follows the PUBLISHED algorithm of that package and is pinned on
mathematics instead of on bytes of the Go binary:
  * the 2-adic root of unity is the one gnark-crypto hard-codes for BN254 Fr,
    19103219067921713944291392827692070036145651957329286315305642004821462161904 = 5^((q-1)/2^28), of order exactly 2^28
    (asserted below);
  * `Domain.Generator` has order n, `Domain.FinerGenerator` order n * 2^depth, FinerGenerator^(2^depth) = Generator;
  * `FFT(a, DIF, 0)` maps coefficients (natural order) to evaluations in bit-reversed order, `FFT(a, DIT, coset)` maps
    bit-reversed input to natural-order evaluations on FinerGenerator^coset * <Generator>, `FFTInverse` uses the inverse
    twiddles and scales by 1/n (and by the inverse coset table);
  * the defining identity: with A, B, C the interpolants of a, b, c on <Generator> and Z = X^n - 1,
    H = (A*B - C) / Z whenever a*b = c pointwise (tests check it with schoolbook polynomial arithmetic), and in general
    computeH's output is the coefficient vector of the degree-< n polynomial that agrees with (A*B - C) * (-2)^-1 on the
    odd coset, IN BIT-REVERSED ORDER (the reference's version of computeH does not bit-reverse after the last
    FFTInverse(DIF)) and in REGULAR form (FromMont, prove.go:352-356).
"""
Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ROOT_2_28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904
MAX_ORDER_ROOT = 28
assert pow(ROOT_2_28, 1 << 28, Q) == 1 and pow(ROOT_2_28, 1 << 27, Q) == Q - 1 and pow(5, (Q - 1) >> 28, Q) == ROOT_2_28


def bit_reverse(i, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


class Domain:
    """fft.NewDomain(m, depth, precomputeReversedTable) of gnark-crypto v0.6.x (ecc/bn254/fr/fft/domain.go)."""

    def __init__(self, m, depth=1):
        x = 1
        while x < m:
            x <<= 1
        self.cardinality = x
        self.depth = depth
        self.log = x.bit_length() - 1
        if self.log + depth > MAX_ORDER_ROOT:
            raise ValueError("m is too big: the required root of unity does not exist")
        self.finer_generator = pow(ROOT_2_28, 1 << (MAX_ORDER_ROOT - (self.log + depth)), Q)
        self.finer_generator_inv = pow(self.finer_generator, Q - 2, Q)
        self.generator = pow(ROOT_2_28, 1 << (MAX_ORDER_ROOT - self.log), Q)
        self.generator_inv = pow(self.generator, Q - 2, Q)
        self.cardinality_inv = pow(x, Q - 2, Q)
        assert pow(self.finer_generator, 1 << depth, Q) == self.generator

    # twiddles[stage][i] = (generator^(2^stage))^i, as precomputeTwiddles lays them out
    def twiddle(self, stage, i, inverse=False):
        g = self.generator_inv if inverse else self.generator
        return pow(g, (1 << stage) * i, Q)

    def coset(self, c, i, inverse=False):      # CosetTable[c-1][i] = (FinerGenerator^c)^i
        g = self.finer_generator_inv if inverse else self.finer_generator
        return pow(g, c * i, Q)


def _dif(a, lo, n, dom, stage, inverse):       # difFFT (fft.go): butterflies, then the two halves
    if n == 1:
        return
    m = n >> 1
    for i in range(m):
        x, y = a[lo + i], a[lo + i + m]
        a[lo + i] = (x + y) % Q
        a[lo + i + m] = (x - y) * dom.twiddle(stage, i, inverse) % Q
    _dif(a, lo, m, dom, stage + 1, inverse)
    _dif(a, lo + m, m, dom, stage + 1, inverse)


def _dit(a, lo, n, dom, stage, inverse):       # ditFFT: the two halves, then the butterflies
    if n == 1:
        return
    m = n >> 1
    _dit(a, lo, m, dom, stage + 1, inverse)
    _dit(a, lo + m, m, dom, stage + 1, inverse)
    for i in range(m):
        x, y = a[lo + i], a[lo + i + m] * dom.twiddle(stage, i, inverse) % Q
        a[lo + i] = (x + y) % Q
        a[lo + i + m] = (x - y) % Q


def fft(dom, a, decimation, coset=0):
    """Domain.FFT (fft.go): in place on a list of ints; decimation 'DIF' (output bit-reversed) or 'DIT' (input bit-reversed)."""
    n = len(a)
    assert n == dom.cardinality
    if coset:
        for i in range(n):
            j = bit_reverse(i, dom.log) if decimation == "DIT" else i     # CosetTableReversed for DIT
            a[i] = a[i] * dom.coset(coset, j) % Q
    (_dif if decimation == "DIF" else _dit)(a, 0, n, dom, 0, False)


def fft_inverse(dom, a, decimation, coset=0):
    """Domain.FFTInverse: inverse twiddles, then CardinalityInv (and the inverse coset table, reversed for DIF)."""
    n = len(a)
    assert n == dom.cardinality
    (_dif if decimation == "DIF" else _dit)(a, 0, n, dom, 0, True)
    for i in range(n):
        f = dom.cardinality_inv
        if coset:
            j = bit_reverse(i, dom.log) if decimation == "DIF" else i     # CosetTableInvReversed for DIF
            f = f * dom.coset(coset, j, inverse=True) % Q
        a[i] = a[i] * f % Q


def compute_h(a, b, c, domain_size=None):
    """computeH(a, b, c, domain) of prover/gadget/prove.go:308-359 on lists of field VALUES (ints).  Returns the list the Go
    function returns, as the values its FromMont leaves in the elements (i.e. the regular-form integers), n = domain size."""
    dom = Domain(domain_size or max(len(a), 1), 1)
    n = dom.cardinality
    a, b, c = (list(v) + [0] * (n - len(v)) for v in (a, b, c))          # :319-323 padding
    for v in (a, b, c):
        fft_inverse(dom, v, "DIF", 0)                                      # :326-328
    for v in (a, b, c):
        fft(dom, v, "DIT", 1)                                              # :330-332
    minus_two_inv = pow(Q - 2, Q - 2, Q)                                   # :334-337
    for i in range(n):                                                     # :341-347
        a[i] = (a[i] * b[i] - c[i]) * minus_two_inv % Q
    fft_inverse(dom, a, "DIF", 1)                                          # :350
    return a                                                               # FromMont: the values themselves


# ---- independent checks (schoolbook; used by the tests) ----------------------------------------------------------
def interpolate_naive(vals, dom):
    """Coefficients (natural order) of the polynomial of degree < n with P(generator^i) = vals[i]: the inverse DFT by its
    definition, O(n^2)."""
    n = dom.cardinality
    return [sum(vals[i] * pow(dom.generator_inv, i * k, Q) for i in range(n)) * dom.cardinality_inv % Q for k in range(n)]


def h_by_division(a, b, c, dom):
    """(A*B - C) / (X^n - 1) by schoolbook multiplication and division; requires a*b = c pointwise (exact division)."""
    n = dom.cardinality
    A, B, C = (interpolate_naive(v, dom) for v in (a, b, c))
    prod = [0] * (2 * n - 1)
    for i, x in enumerate(A):
        if x:
            for j, y in enumerate(B):
                prod[i + j] = (prod[i + j] + x * y) % Q
    for i, x in enumerate(C):
        prod[i] = (prod[i] - x) % Q
    # divide by X^n - 1: prod = H * X^n - H  =>  H_k = prod[n + k] for the top part, and the remainder must vanish
    H = [prod[n + k] if n + k < len(prod) else 0 for k in range(n)]
    for k in range(n):
        assert (prod[k] + H[k]) % Q == 0, "A*B - C is not divisible by X^n - 1"
    return H
