"""GPU parity tests of the G2 multi-scalar multiplication ((*G2Jac).MultiExp at prover/gadget/prove.go:277, Bs over pk.G2.B)
through the C ABI: bit-exact (affine images) against the big-integer oracle of oracle/pyoracle_ec.py ("parity unpinned" against
Go bytes: gnark-crypto is un-vendored; pinned by tests/test_oracle_ec.py on the twist equation, the generator, [r] G2 = infinity
and the group axioms).  Sizes the Python oracle does not reach in seconds are checked through the bases' known discrete
logarithms: MSM over [k_i] G2 with scalars s_i must be [sum k_i s_i] G2 -- one scalar multiplication for the oracle."""
import importlib
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import coracle as c  # noqa: E402
import pyoracle_ec as ec  # noqa: E402
from test_gpu_msm import rand_scalars, _synth_scalars  # noqa: E402

pytestmark = pytest.mark.gpu
Q = ec.R_ORDER
G2IMG = ec.g2_point_to_image(ec.G2)


@pytest.fixture(scope="module")
def gk():
    g = importlib.import_module("gkr-mimc_amd")
    g.init(0)
    return g


def ints(arr):
    return [sum(int(row[k]) << (64 * k) for k in range(4)) for row in np.asarray(arr).reshape(-1, 4)]


def test_g2_generator_and_batch_scalar_mul(gk):
    assert gk.g2_generator().tolist() == G2IMG.tolist()
    rng = random.Random(3)
    sc = rand_scalars(rng, 24)
    got = gk.batch_scalar_multiplication_g2(G2IMG, sc)
    for row, k in zip(got, ints(sc)):
        assert row.tolist() == ec.g2_point_to_image(ec.g2_mul(k, ec.G2)).tolist(), k
    assert not gk.batch_scalar_multiplication_g2(np.zeros(16, dtype=np.uint64), sc[:3]).any()


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 100])
def test_msm_g2_vs_oracle(gk, n):
    rng = random.Random(200 + n)
    ks = [rng.randrange(Q) for _ in range(n)]
    pts = [ec.g2_mul(k, ec.G2) for k in ks]
    if n >= 17:
        pts[3] = ec.INF                     # gnark-crypto's (0, 0): skipped
        pts[5] = pts[4]                     # repeated point
        pts[11] = ec.g2_neg(pts[10])        # a point and its negative
    img = ec.g2_points_to_image(pts)
    sc = rand_scalars(rng, n)
    if n >= 17:
        sc[5] = sc[4]                       # same point, same scalar: the bucket meets P + P (doubling branch)
        sc[11] = sc[10]                     # P and -P in one bucket
    want = ec.g2_point_to_image(ec.g2_msm(pts, ints(sc)))
    assert gk.multi_exp_g2(img, sc).tolist() == want.tolist()
    if n:
        b = gk.G2Bases(points=img)
        assert len(b) == n and b.multi_exp(sc).tolist() == want.tolist()
        m = n // 2
        assert b.multi_exp(sc[:m]).tolist() == ec.g2_point_to_image(ec.g2_msm(pts[:m], ints(sc[:m]))).tolist()
        for cw in (9, 16, 22):              # fixed-base tables over Fp2 (gkrhip_msm_g2_precompute): the same image
            b.precompute(cw)
            assert b.multi_exp(sc).tolist() == want.tolist(), cw
            assert b.multi_exp(sc[:m]).tolist() == ec.g2_point_to_image(ec.g2_msm(pts[:m], ints(sc[:m]))).tolist(), cw
        b.precompute(-1)
        assert b.multi_exp(sc).tolist() == want.tolist()
        b.close()


def _dlog_check(gk, n, seed, cw=0, skew=False):
    """bases [k_i] G2 generated on the device, MSM(s) against [sum k_i s_i] G2 (the oracle's single scalar multiplication)."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    b = gk.G2Bases(base=G2IMG, scalars=k)
    if cw:
        b.set_window(cw)
    if skew:
        r = random.Random(seed)
        s = ec.scalars_to_image([r.choice([0, 1, 1, 1, 2, Q - 1]) for _ in range(n)])
    else:
        s = rand_scalars(random.Random(seed), n)
    tot = sum(x * y for x, y in zip(ints(k), ints(s))) % Q
    assert b.multi_exp(s).tolist() == ec.g2_point_to_image(ec.g2_mul(tot, ec.G2)).tolist(), (n, seed, cw, skew)
    head = b.read(0, 2)
    assert head[1].tolist() == ec.g2_point_to_image(ec.g2_mul(ints(k[1:2])[0], ec.G2)).tolist()
    b.close()


@pytest.mark.parametrize("cw", [2, 5, 8, 11, 13, 16])
def test_msm_g2_window_sizes(gk, cw):
    _dlog_check(gk, 700, 40 + cw, cw=cw)


def test_msm_g2_larger_sizes_and_skew(gk):
    _dlog_check(gk, 1 << 12, 1)
    _dlog_check(gk, 1 << 16, 2)
    _dlog_check(gk, 5000, 3, skew=True)          # the 0/1 wires of a real witness: the workgroup-per-bucket path
    _dlog_check(gk, 1 << 15, 4, skew=True)       # ... with buckets of several segments
    _dlog_check(gk, 1 << 14, 5, cw=14)           # a top window of two bits: four buckets of n / 4 points
    # all-cancelling input: the point at infinity, encoded as zeros
    k = rand_scalars(random.Random(9), 50)
    pts = gk.batch_scalar_multiplication_g2(G2IMG, k)
    pm = np.concatenate([pts, pts])
    sc = ec.scalars_to_image([7] * 50 + [Q - 7] * 50)
    assert not gk.multi_exp_g2(pm, sc).any()


def test_msm_g2_errors_and_montgomery_scalars(gk):
    rng = random.Random(5)
    k = rand_scalars(rng, 6)
    pts = gk.batch_scalar_multiplication_g2(G2IMG, k)
    bad = pts.copy()
    bad[2, 7] = np.uint64(0xFFFFFFFFFFFFFFFF)            # X.A1 >= p
    with pytest.raises(gk.GkrHipError, match="canonical"):
        gk.multi_exp_g2(bad, k)
    vals = [rng.randrange(Q) for _ in range(6)]
    want = gk.multi_exp_g2(pts, ec.scalars_to_image(vals))
    assert gk.multi_exp_g2(pts, c.from_ints(vals), scalars_mont=True).tolist() == want.tolist()
    assert want.tolist() == ec.g2_point_to_image(ec.g2_mul(sum(a * b for a, b in zip(ints(k), vals)) % Q, ec.G2)).tolist()


def test_bench_msm_g2_result(gk):
    """The G2 micro-benchmark computes a real MSM (same synthetic scalars as the G1 one, bases [k_i] G2)."""
    logn = 8
    n = 1 << logn
    r = gk.bench_msm_g2(logn, warmup=1, iters=2)
    tot = sum(a * b for a, b in zip(_synth_scalars(n, 0x1234567), _synth_scalars(n, 0x7654321))) % Q
    assert r["result"].tolist() == ec.g2_point_to_image(ec.g2_mul(tot, ec.G2)).tolist()
    assert r["ms"] > 0


def test_msm_g2_two_level_sort(gk):
    """The two-level sort (default from 2^20 points) forced at sizes the oracle reaches: the sorting kernels are shared with G1."""
    gk.set_option("msm_sort_levels", 2)
    try:
        _dlog_check(gk, 700, 61)
        _dlog_check(gk, 1 << 14, 62)
        _dlog_check(gk, 1 << 15, 63, skew=True)
        _dlog_check(gk, 1 << 14, 64, cw=14)
    finally:
        gk.set_option("msm_sort_levels", 0)


def test_msm_g1_g2_pair_shares_the_sort(gk):
    """bs1 and Bs of prove.go:189,277 run over the same scalars: gkrhip_msm_g1_g2 gives exactly the two separate results (oracle
    at a small size, the separate calls at 2^16 and with skewed scalars; prefixes; mismatched handles are refused)."""
    import coracle as c
    for n, seed, skew in ((0, 1, False), (1, 2, False), (300, 3, False), (1 << 16, 4, False), (1 << 15, 5, True)):
        rng = np.random.default_rng(seed)
        k = rng.integers(0, 1 << 63, size=(max(n, 1), 4), dtype=np.uint64)
        k[:, 3] &= np.uint64((1 << 60) - 1)
        b1 = gk.G1Bases(base=c.G1_GEN, scalars=k)
        b2 = gk.G2Bases(base=G2IMG, scalars=k)
        if skew:
            r = random.Random(seed)
            s = ec.scalars_to_image([r.choice([0, 1, 1, 1, 2, Q - 1]) for _ in range(n)])
        else:
            s = rand_scalars(random.Random(seed), n)
        o1, o2 = gk.multi_exp_g1_g2(b1, b2, s[:n])
        assert o1.tolist() == b1.multi_exp(s[:n]).tolist() and o2.tolist() == b2.multi_exp(s[:n]).tolist(), n
        if n in (300, 1 << 15):
            # both handles on fixed-base tables of one geometry: the tables' sort once, each handle's sums on its own tables; tables of
            # different geometries (or on one handle only): the per-window path, the same points
            for c1, c2 in ((13, 13), (20, 20), (13, 16), (0, -1)):
                b1.precompute(c1)
                b2.precompute(c2)
                f1, f2 = gk.multi_exp_g1_g2(b1, b2, s[:n])
                assert f1.tolist() == o1.tolist() and f2.tolist() == o2.tolist(), (n, c1, c2)
                m = n // 3
                h1, h2 = gk.multi_exp_g1_g2(b1, b2, s[:m])
                assert h1.tolist() == b1.multi_exp(s[:m]).tolist() and h2.tolist() == b2.multi_exp(s[:m]).tolist(), (n, c1, c2)
            b1.precompute(-1)
            b2.precompute(-1)
        if n == 300:
            assert o1.tolist() == c.g1_msm(b1.read(0, n), s).tolist()
            tot = sum(x * y for x, y in zip(ints(k), ints(s))) % Q
            assert o2.tolist() == ec.g2_point_to_image(ec.g2_mul(tot, ec.G2)).tolist()
            h1, h2 = gk.multi_exp_g1_g2(b1, b2, s[:100])
            assert h1.tolist() == b1.multi_exp(s[:100]).tolist() and h2.tolist() == b2.multi_exp(s[:100]).tolist()
            b1.set_window(11)                       # the G1 handle's window serves both
            p1, p2 = gk.multi_exp_g1_g2(b1, b2, s)
            assert p1.tolist() == o1.tolist() and p2.tolist() == o2.tolist()
            b3 = gk.G2Bases(base=G2IMG, scalars=k[:200])
            with pytest.raises(gk.GkrHipError):
                gk.multi_exp_g1_g2(b1, b3, s[:100])
            bad = s.copy()
            bad[7] = np.uint64(0xFFFFFFFFFFFFFFFF)
            with pytest.raises(gk.GkrHipError):
                gk.multi_exp_g1_g2(b1, b2, bad)
            b3.close()
        b1.close()
        b2.close()


def test_msm_shared_over_unfiltered_wires(gk):
    """gkrhip_msm_shared: ar, bs1 (G1) and Bs (G2) over the UNFILTERED wire vector, the key's vectors expanded with points at
    infinity where pk.InfinityA / pk.InfinityB drop wires (prove.go:136-160): one sort serves the three sums, and each equals
    the reference's filtered MSM."""
    import coracle as c
    n = 3000
    rng = np.random.default_rng(90)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    pa = c.g1_batch_scalar_mul(c.G1_GEN, k)
    pb1 = pa.copy()
    pb2 = gk.batch_scalar_multiplication_g2(G2IMG, k)
    inf_a = rng.random(n) < 0.1
    inf_b = rng.random(n) < 0.3
    pa[inf_a] = 0
    pb1[inf_b] = 0
    pb2[inf_b] = 0
    wires = rand_scalars(random.Random(90), n)
    ba, bb1, bb2 = gk.G1Bases(points=pa), gk.G1Bases(points=pb1), gk.G2Bases(points=pb2)
    (ar, bs1), (bs2,) = gk.multi_exp_shared([ba, bb1], [bb2], wires)
    # the reference's calls: filtered bases, filtered scalars
    assert ar.tolist() == c.g1_msm(pa[~inf_a], wires[~inf_a]).tolist()
    assert bs1.tolist() == c.g1_msm(pb1[~inf_b], wires[~inf_b]).tolist()
    tot = sum(x * y for x, y, dead in zip(ints(k), ints(wires), inf_b) if not dead) % Q
    assert bs2.tolist() == ec.g2_point_to_image(ec.g2_mul(tot, ec.G2)).tolist()
    assert bs2.tolist() == bb2.multi_exp(wires).tolist()
    # G1 only, G2 only, nothing; a handle twice and a G2 handle in the G1 list are refused
    (x,), none = gk.multi_exp_shared([ba], [], wires)
    assert x.tolist() == ar.tolist() and none == []
    none, (y,) = gk.multi_exp_shared([], [bb2], wires[:100])
    assert y.tolist() == bb2.multi_exp(wires[:100]).tolist()
    assert gk.multi_exp_shared([], [], wires) == ([], [])
    with pytest.raises(gk.GkrHipError, match="twice"):
        gk.multi_exp_shared([ba, ba], [], wires)
    with pytest.raises(gk.GkrHipError, match="not a G1 handle"):
        gk.multi_exp_shared([bb2], [], wires)
    for b in (ba, bb1, bb2):
        b.close()
