"""Big-integer oracle of BN254 G1 and of the multi-scalar multiplication.  TEST INFRASTRUCTURE ONLY: nothing under
gkr-mimc_amd/ imports, links or executes this file.

What it stands for: gnark-crypto's `(*G1Jac).MultiExp` / `(*G1Affine).MultiExp` and `BatchScalarMultiplicationG1` as the
reference calls them at prover/gadget/prove.go:76,91,177,189,202,221.  gnark-crypto (v0.6.1-0.20220110145513-493bb1c180d9,
go.mod:7) is NOT under /root/reference, so there is no vector of the reference to pin against: **parity unpinned** with
respect to bytes of the Go binary.  What pins it instead is the mathematics: the result of an MSM is a group element whose
affine coordinates are unique, and this file computes it with the textbook affine chord-and-tangent law on Python integers
(no projective coordinates, no windows, no Montgomery form) -- checked against the curve equation y^2 = x^3 + 3, the
generator (1, 2) (gnark-crypto's bn254 g1Gen), [r] G = infinity for the group order r = q (the Fr modulus of hash/ark.go:7)
and the group axioms in tests/test_oracle_ec.py.

Memory images (what the C ABI exchanges) are gnark-crypto's: fp.Element = 4 little-endian uint64 limbs of x * 2^256 mod p;
G1Affine = {X, Y}; the point at infinity is (0, 0).
"""
from __future__ import annotations

import numpy as np

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583      # base field
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # group order = Fr modulus
B = 3
G = (1, 2)
INF = None
MONT_R = (1 << 256) % P
MONT_RINV = pow(MONT_R, -1, P)


def on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def neg(pt):
    return INF if pt is INF else (pt[0], (-pt[1]) % P)


def add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def mul(k, pt):
    """[k] pt by right-to-left double-and-add (k any non-negative integer)."""
    acc = INF
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def msm(points, scalars):
    acc = INF
    for p, s in zip(points, scalars):
        acc = add(acc, mul(s, p))
    return acc


# ---- memory images ---------------------------------------------------------------------------------------------
def fp_to_limbs(x_regular):
    v = x_regular * MONT_R % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def fp_from_limbs(l):
    v = sum(int(l[i]) << (64 * i) for i in range(4))
    return v * MONT_RINV % P


def point_to_image(pt):
    """(8,) uint64: gnark-crypto's G1Affine (Montgomery X, Y; infinity = zeros)."""
    if pt is INF:
        return np.zeros(8, dtype=np.uint64)
    return np.array(fp_to_limbs(pt[0]) + fp_to_limbs(pt[1]), dtype=np.uint64)


def point_from_image(img):
    img = np.asarray(img, dtype=np.uint64).reshape(8)
    if not img.any():
        return INF
    return (fp_from_limbs(img[:4]), fp_from_limbs(img[4:]))


def points_to_image(pts):
    return np.stack([point_to_image(p) for p in pts]) if len(pts) else np.zeros((0, 8), dtype=np.uint64)


def scalar_to_limbs(s):
    """(4,) uint64: the REGULAR-form image of a scalar (what fr.Element holds after FromMont)."""
    return np.array([(s >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def scalars_to_image(ss):
    return np.stack([scalar_to_limbs(s) for s in ss]) if len(ss) else np.zeros((0, 4), dtype=np.uint64)


def scalar_from_limbs(l):
    return sum(int(l[i]) << (64 * i) for i in range(4))


# ---- G2: the same textbook law over Fp2 = Fp[u] / (u^2 + 1) on the twist y^2 = x^3 + 3 / (9 + u) ---------------------------
# (gnark-crypto's bn254.G2Affine {X, Y fptower.E2{A0, A1}}; MultiExp on G2 at prover/gadget/prove.go:277.)  Elements of Fp2 are
# pairs (a0, a1) of Python integers.  Pinned in tests/test_oracle_ec.py: generator on the twist, [r] G2 = infinity, group axioms.
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1] * n) % P)


def f2_small(k):
    return (k % P, 0)


B2 = f2_mul(f2_small(3), f2_inv((9, 1)))
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))      # gnark-crypto's g2Gen


def g2_on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_neg(pt):
    return INF if pt is INF else (pt[0], f2_sub((0, 0), pt[1]))


def g2_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return INF
        lam = f2_mul(f2_mul(f2_small(3), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(k, pt):
    acc = INF
    while k:
        if k & 1:
            acc = g2_add(acc, pt)
        pt = g2_add(pt, pt)
        k >>= 1
    return acc


def g2_msm(points, scalars):
    acc = INF
    for p, s in zip(points, scalars):
        acc = g2_add(acc, g2_mul(s, p))
    return acc


def g2_point_to_image(pt):
    """(16,) uint64: gnark-crypto's G2Affine {X.A0, X.A1, Y.A0, Y.A1} (Montgomery; infinity = zeros)."""
    if pt is INF:
        return np.zeros(16, dtype=np.uint64)
    (x0, x1), (y0, y1) = pt
    return np.array(fp_to_limbs(x0) + fp_to_limbs(x1) + fp_to_limbs(y0) + fp_to_limbs(y1), dtype=np.uint64)


def g2_point_from_image(img):
    img = np.asarray(img, dtype=np.uint64).reshape(16)
    if not img.any():
        return INF
    return ((fp_from_limbs(img[0:4]), fp_from_limbs(img[4:8])), (fp_from_limbs(img[8:12]), fp_from_limbs(img[12:16])))


def g2_points_to_image(pts):
    return np.stack([g2_point_to_image(p) for p in pts]) if len(pts) else np.zeros((0, 16), dtype=np.uint64)
