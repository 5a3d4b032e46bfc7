"""CPU tests that pin the G1 oracles (oracle/pyoracle_ec.py big-integer affine arithmetic, oracle/g1_oracle.c Jacobian
double-and-add) before the GPU multi-scalar multiplication is checked against them.  gnark-crypto is an un-vendored
dependency of the reference (go.mod:7), so no vector of the reference exists for prove.go's MultiExp calls: PARITY UNPINNED
against Go bytes; pinned here on the curve equation, the generator, the group order, published multiples of G and the group
axioms, and the two independent implementations against each other.  Also: the product's host-side Fp / G1 code and the
generated Fp schedules on the CPU (tests/cpp/test_fp_g1_host.cpp)."""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import coracle as c  # noqa: E402
import pyoracle_ec as ec  # noqa: E402

# [2]G and [3]G of BN254 G1 (alt_bn128; the constants of EIP-196's test vectors)
G2 = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
      9918110051302171585080402603319702774565515993150576347155970296011118125764)
G3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
      19321533766552368860946552437480515441416830039777911637913418824951667761761)


def test_curve_facts_python():
    assert ec.on_curve(ec.G) and ec.G == (1, 2)
    assert ec.add(ec.G, ec.G) == G2 and ec.add(G2, ec.G) == G3 and ec.mul(3, ec.G) == G3
    assert ec.mul(ec.R_ORDER, ec.G) is ec.INF                     # the group order is the Fr modulus (hash/ark.go:7)
    assert ec.mul(ec.R_ORDER - 1, ec.G) == ec.neg(ec.G)
    assert ec.add(ec.G, ec.neg(ec.G)) is ec.INF and ec.add(ec.INF, ec.G) == ec.G
    rng = random.Random(5)
    a, b = rng.randrange(ec.R_ORDER), rng.randrange(ec.R_ORDER)
    pa, pb = ec.mul(a, ec.G), ec.mul(b, ec.G)
    assert ec.on_curve(pa) and ec.add(pa, pb) == ec.mul((a + b) % ec.R_ORDER, ec.G) == ec.add(pb, pa)
    assert ec.mul(b, pa) == ec.mul(a * b % ec.R_ORDER, ec.G)


def test_images():
    assert ec.point_to_image(ec.G).tolist() == c.G1_GEN.tolist()
    assert ec.point_from_image(ec.point_to_image(G3)) == G3
    assert ec.point_from_image(np.zeros(8, dtype=np.uint64)) is ec.INF
    assert ec.scalar_from_limbs(ec.scalar_to_limbs(ec.R_ORDER - 1)) == ec.R_ORDER - 1


def test_c_oracle_scalar_mul_vs_python():
    rng = random.Random(11)
    ks = [0, 1, 2, 3, 7, 2 ** 64 - 1, 2 ** 64, 2 ** 128 + 5, ec.R_ORDER - 1, ec.R_ORDER, ec.R_ORDER + 1] + [rng.randrange(ec.R_ORDER) for _ in range(12)]
    base = ec.mul(rng.randrange(ec.R_ORDER), ec.G)
    for k in ks:
        for b in (ec.G, base):
            got = c.g1_scalar_mul(ec.point_to_image(b), ec.scalar_to_limbs(k))
            assert got.tolist() == ec.point_to_image(ec.mul(k, b)).tolist(), k
            assert c.g1_on_curve(got)
    assert not c.g1_on_curve(np.array([1, 0, 0, 0, 1, 0, 0, 0], dtype=np.uint64))
    inf = np.zeros(8, dtype=np.uint64)
    assert c.g1_scalar_mul(inf, ec.scalar_to_limbs(5)).tolist() == inf.tolist()


def test_c_oracle_add_special_cases():
    rng = random.Random(12)
    p = ec.mul(rng.randrange(ec.R_ORDER), ec.G)
    q = ec.mul(rng.randrange(ec.R_ORDER), ec.G)
    for a, b in ((p, q), (p, p), (p, ec.neg(p)), (ec.INF, q), (p, ec.INF), (ec.INF, ec.INF)):
        assert c.g1_add(ec.point_to_image(a), ec.point_to_image(b)).tolist() == ec.point_to_image(ec.add(a, b)).tolist()


def test_c_oracle_msm_vs_python():
    rng = random.Random(13)
    for n in (0, 1, 2, 5, 33):
        pts = [ec.mul(rng.randrange(ec.R_ORDER), ec.G) for _ in range(n)]
        ss = [rng.randrange(ec.R_ORDER) for _ in range(n)]
        if n >= 5:
            pts[1] = ec.INF                 # gnark-crypto's (0, 0): skipped
            ss[2] = 0
            ss[3] = ec.R_ORDER - 1
            pts[4] = pts[0]                 # repeated point
        got = c.g1_msm(ec.points_to_image(pts), ec.scalars_to_image(ss))
        assert got.tolist() == ec.point_to_image(ec.msm(pts, ss)).tolist(), n


def test_c_oracle_batch_and_linearity():
    rng = np.random.default_rng(3)
    n = 256
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    pts = c.g1_batch_scalar_mul(c.G1_GEN, sc)
    assert all(c.g1_on_curve(p) for p in pts[:16])
    assert pts[7].tolist() == c.g1_scalar_mul(c.G1_GEN, sc[7]).tolist()
    # sum_i [k_i] ([s_i] G) == [sum k_i s_i] G
    ks = [int(x) for x in rng.integers(0, 1 << 62, size=n)]
    tot = sum(k * ec.scalar_from_limbs(s) for k, s in zip(ks, sc)) % ec.R_ORDER
    got = c.g1_msm(pts, ec.scalars_to_image(ks))
    assert got.tolist() == c.g1_scalar_mul(c.G1_GEN, ec.scalar_to_limbs(tot)).tolist()


def test_g2_curve_facts_python():
    """G2 of the oracle: gnark-crypto's generator on the twist y^2 = x^3 + 3/(9 + u), order r, group axioms, images."""
    assert ec.g2_on_curve(ec.G2)
    assert ec.g2_mul(ec.R_ORDER, ec.G2) is ec.INF
    assert ec.g2_mul(ec.R_ORDER - 1, ec.G2) == ec.g2_neg(ec.G2)
    rng = random.Random(6)
    a, b = rng.randrange(ec.R_ORDER), rng.randrange(ec.R_ORDER)
    pa, pb = ec.g2_mul(a, ec.G2), ec.g2_mul(b, ec.G2)
    assert ec.g2_on_curve(pa) and ec.g2_add(pa, pb) == ec.g2_mul((a + b) % ec.R_ORDER, ec.G2) == ec.g2_add(pb, pa)
    assert ec.g2_mul(b, pa) == ec.g2_mul(a * b % ec.R_ORDER, ec.G2)
    assert ec.g2_add(pa, ec.g2_neg(pa)) is ec.INF and ec.g2_add(pa, pa) == ec.g2_mul(2, pa)
    assert ec.g2_point_from_image(ec.g2_point_to_image(pa)) == pa
    assert ec.g2_msm([pa, pb, ec.INF], [3, 5, 7]) == ec.g2_mul((3 * a + 5 * b) % ec.R_ORDER, ec.G2)


def test_fp_schedule_and_host_g1_on_cpu():
    """The generated Fp schedules (portable branch, every untracked multiply-add checked for wrap-around) and the host curve
    code of the MSM's scalar tail against the oracle."""
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "t")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_fp_g1_host.cpp"),
                               "-L" + os.path.join(ROOT, "oracle"), "-lgkr_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                               "-Wl,-rpath,/opt/rocm/lib/llvm/lib"])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "bad=0 skip_overflows=0" in out.stdout
