"""GPU parity tests of the G1 multi-scalar multiplication (gnark-crypto's MultiExp as the reference calls it at
prover/gadget/prove.go:76,91,189,202,221; BatchScalarMultiplicationG1 at :177) through the C ABI:
bit-exact (affine images) against the C oracle's per-term double-and-add (oracle/g1_oracle.c, itself pinned on big-integer
arithmetic by tests/test_oracle_ec.py -- "parity unpinned" against Go bytes: un-vendored dependency), and at sizes the
oracle does not reach in seconds through the linearity of the map scalars -> MSM."""
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

pytestmark = pytest.mark.gpu
Q = ec.R_ORDER


@pytest.fixture(scope="module")
def gk():
    g = importlib.import_module("gkr-mimc_amd")
    g.init(0)
    return g


def rand_scalars(rng, n):
    """(n, 4) regular-form scalars below q, with the corner values the window recoding cares about."""
    vals = [rng.randrange(Q) for _ in range(n)]
    corners = [0, 1, Q - 1, Q - 2, 2 ** 253, 2 ** 254 - 1 if 2 ** 254 - 1 < Q else Q - 3, 0x8000, 0x7fff, 0x8001, 0xffff, 0x10000,
               (1 << 128) - 1, 1 << 128, int("8000" * 15, 16), int("7fff" * 15, 16), int("ffff" * 15, 16) % Q]
    for i, v in enumerate(corners):
        if i < n:
            vals[(7 * i + 3) % n] = v % Q
    return ec.scalars_to_image(vals) if n else np.zeros((0, 4), dtype=np.uint64)


def rand_points(seed, n):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    return c.g1_batch_scalar_mul(c.G1_GEN, k) if n else np.zeros((0, 8), dtype=np.uint64)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 17, 64, 100, 1000, 4096, 1 << 14])
def test_msm_vs_oracle(gk, n):
    rng = random.Random(100 + n)
    pts = rand_points(n + 1, n)
    if n >= 17:
        pts[3] = 0                      # point at infinity, gnark-crypto's (0, 0)
        pts[5] = pts[4]                 # repeated point
        pts[9] = pts[4]
        pts[11, 4:] = ec.point_to_image(ec.neg(ec.point_from_image(pts[10])))[4:]   # a point and its negative
        pts[11, :4] = pts[10, :4]
    sc = rand_scalars(rng, n)
    if n >= 17:
        sc[5] = sc[4]                   # same point, same scalar: the bucket meets P + P (doubling branch)
        sc[11] = sc[10]                 # P and -P in one bucket: the sum passes through infinity
    want = c.g1_msm(pts, sc)
    assert gk.multi_exp_g1(pts, sc).tolist() == want.tolist()
    if n:
        b = gk.G1Bases(points=pts)
        assert len(b) == n
        assert b.multi_exp(sc).tolist() == want.tolist()
        assert b.multi_exp(sc).tolist() == want.tolist()               # the work buffers are reused
        # a prefix of the bases (prove.go's MSMs run over filtered sub-vectors)
        m = n // 2
        assert b.multi_exp(sc[:m]).tolist() == c.g1_msm(pts[:m], sc[:m]).tolist()
        b.close()


@pytest.mark.parametrize("cw", list(range(2, 17)))
def test_msm_every_window_size(gk, cw):
    """Every window size gives the same point (the recoding's carries, the top window, the chunked reduction)."""
    rng = random.Random(cw)
    n = 300
    pts = rand_points(77, n)
    sc = rand_scalars(rng, n)
    b = gk.G1Bases(points=pts)
    b.set_window(cw)
    assert b.multi_exp(sc).tolist() == c.g1_msm(pts, sc).tolist()
    b.close()


def test_msm_skewed_scalars(gk):
    """The 0/1 wires of a real witness: most points of window 0 land in one bucket (the workgroup-per-bucket path), and a
    bucket holding one point many times exercises the doubling branch of the mixed addition."""
    n = 6000
    pts = rand_points(5, n)
    ones = ec.scalars_to_image([1] * n)
    assert gk.multi_exp_g1(pts, ones).tolist() == c.g1_msm(pts, ones).tolist()
    rng = random.Random(9)
    sc = ec.scalars_to_image([rng.choice([0, 1, 1, 1, 2, Q - 1]) for _ in range(n)])
    assert gk.multi_exp_g1(pts, sc).tolist() == c.g1_msm(pts, sc).tolist()
    same = np.repeat(pts[:1], n, axis=0)           # n copies of one point, equal scalars: [n * s] P
    s = rng.randrange(Q)
    want = c.g1_scalar_mul(pts[0], ec.scalar_to_limbs(n * s % Q))
    assert gk.multi_exp_g1(same, ec.scalars_to_image([s] * n)).tolist() == want.tolist()
    # everything cancels: the result is the point at infinity, encoded (0, 0)
    pm = np.concatenate([pts[:100], pts[:100]])
    scm = ec.scalars_to_image([5] * 100 + [Q - 5] * 100)
    assert gk.multi_exp_g1(pm, scm).tolist() == [0] * 8


def test_msm_huge_buckets_are_split(gk):
    """A witness-like scalar vector at a size where one bucket holds a large share of all points (2^17 points, 70 % of the
    scalars equal to 1): the bucket is cut into segments, one workgroup each, and the partial sums combined; and a window
    size whose top window is two bits wide (four buckets of n / 4 points)."""
    n = 1 << 17
    pts = rand_points(17, n)
    rng = random.Random(17)
    sc = ec.scalars_to_image([1 if rng.random() < 0.7 else rng.choice([0, 2, 3, Q - 1, rng.randrange(Q)]) for _ in range(n)])
    want = c.g1_msm(pts, sc)
    b = gk.G1Bases(points=pts)
    assert b.multi_exp(sc).tolist() == want.tolist()
    full = rand_scalars(rng, n)
    want_full = c.g1_msm(pts, full)
    for cw in (14, 11):                        # 254 = 18 * 14 + 2 = 23 * 11 + 1: a top window of two bits / one bit
        b.set_window(cw)
        assert b.multi_exp(full).tolist() == want_full.tolist(), cw
        assert b.multi_exp(sc).tolist() == want.tolist(), cw
    b.close()


def test_msm_montgomery_scalars(gk):
    """MultiExpConfig.ScalarsMont: the same scalars handed over as fr.Elements (Montgomery form)."""
    rng = random.Random(21)
    n = 500
    pts = rand_points(8, n)
    vals = [rng.randrange(Q) for _ in range(n)]
    want = c.g1_msm(pts, ec.scalars_to_image(vals))
    assert gk.multi_exp_g1(pts, c.from_ints(vals), scalars_mont=True).tolist() == want.tolist()


def test_batch_scalar_multiplication(gk):
    """bn254.BatchScalarMultiplicationG1 (prove.go:177) and the device-generated bases."""
    rng = random.Random(31)
    base = rand_points(9, 1)[0]
    sc = rand_scalars(rng, 200)
    want = c.g1_batch_scalar_mul(base, sc)
    assert np.array_equal(gk.batch_scalar_multiplication_g1(base, sc), want)
    b = gk.G1Bases(base=base, scalars=sc)
    assert np.array_equal(b.read(), want)
    assert np.array_equal(b.read(3, 5), want[3:8])
    k = rand_scalars(rng, 200)
    assert b.multi_exp(k).tolist() == c.g1_msm(want, k).tolist()
    b.close()
    inf = np.zeros(8, dtype=np.uint64)
    assert not gk.batch_scalar_multiplication_g1(inf, sc[:4]).any()


def test_msm_errors(gk):
    pts = rand_points(3, 4)
    bad = pts.copy()
    bad[2, 3] = np.uint64(0xFFFFFFFFFFFFFFFF)          # X >= p: not a canonical fp.Element
    with pytest.raises(gk.GkrHipError, match="canonical"):
        gk.multi_exp_g1(bad, rand_scalars(random.Random(1), 4))
    b = gk.G1Bases(points=pts)
    with pytest.raises(gk.GkrHipError, match="scalars for"):
        b.multi_exp(rand_scalars(random.Random(1), 5))
    with pytest.raises(gk.GkrHipError, match="window"):
        b.set_window(17)
    big = rand_scalars(random.Random(2), 4)
    big[1, 3] = np.uint64(0xFFFFFFFFFFFFFFFF)           # far above 2^254: not a reduced fr.Element
    with pytest.raises(gk.GkrHipError, match="2\\^254"):
        b.multi_exp(big)
    assert b.multi_exp(big[:1]).tolist() == c.g1_msm(pts[:1], big[:1]).tolist()      # the handle is still usable
    b.close()


def test_msm_linearity_2p20(gk):
    """n = 2^20 (the oracle would need minutes): MSM(s) + MSM(t) == MSM(s + t mod q) on device-generated bases, a prefix sum
    against the oracle, and the all-ones MSM against the plain sum of a slice."""
    n = 1 << 20
    rng = np.random.default_rng(2024)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    b = gk.G1Bases(base=c.G1_GEN, scalars=k)
    head = b.read(0, 64)
    assert np.array_equal(head, c.g1_batch_scalar_mul(c.G1_GEN, k[:64]))
    s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    t = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    t[:, 3] &= np.uint64((1 << 60) - 1)
    # s + t as 256-bit integers (both below 2^252: no reduction needed, the sum stays below q)
    st = np.zeros_like(s)
    carry = np.zeros(n, dtype=np.uint64)
    for j in range(4):
        a = s[:, j] + t[:, j]
        c1 = (a < s[:, j]).astype(np.uint64)
        a2 = a + carry
        c2 = (a2 < a).astype(np.uint64)
        st[:, j] = a2
        carry = c1 + c2
    assert not carry.any()
    ms, mt, mst = b.multi_exp(s), b.multi_exp(t), b.multi_exp(st)
    assert c.g1_on_curve(ms) and c.g1_on_curve(mt)
    assert c.g1_add(ms, mt).tolist() == mst.tolist()
    m = 1 << 12
    assert b.multi_exp(s[:m]).tolist() == c.g1_msm(b.read(0, m), s[:m]).tolist()
    # against [sum k_i s_i] G on the whole vector: the bases are known multiples of G
    tot = 0
    ki = [sum(int(row[j]) << (64 * j) for j in range(4)) for row in k[:m]]
    si = [sum(int(row[j]) << (64 * j) for j in range(4)) for row in s[:m]]
    tot = sum(x * y for x, y in zip(ki, si)) % Q
    assert b.multi_exp(s[:m]).tolist() == c.g1_scalar_mul(c.G1_GEN, ec.scalar_to_limbs(tot)).tolist()
    b.close()


def _synth_scalars(n, seed):
    """Python mirror of k_msm_synth_scalars: limbs of (mix(i, seed))^7 as Montgomery products (x^7 R^-6 mod q)."""
    M = (1 << 32) - 1
    rinv = pow(1 << 256, -1, Q)
    out = []
    for i in range(n):
        limbs = [(i ^ 0x9df123f) & M, ((i >> 32) + 0xf45c) & M, seed, 0x2545f491, (i * 0x9e3779b9) & M, 3, seed ^ 0x5bd1e995, 0]
        x = sum(v << (32 * j) for j, v in enumerate(limbs))
        out.append(pow(x, 7, Q) * pow(rinv, 6, Q) % Q)
    return out


def test_bench_msm_result(gk):
    """The micro-benchmark computes a real MSM: its result on the synthetic device-resident data equals the oracle's on the
    same data rebuilt in Python."""
    logn = 9
    n = 1 << logn
    r = gk.bench_msm_g1(logn, warmup=1, iters=2)
    ks = _synth_scalars(n, 0x1234567)
    ss = _synth_scalars(n, 0x7654321)
    tot = sum(a * b for a, b in zip(ks, ss)) % Q
    assert r["result"].tolist() == c.g1_scalar_mul(c.G1_GEN, ec.scalar_to_limbs(tot)).tolist()
    assert r["ms"] > 0 and r["c"] >= 2 and abs(sum(r["phases_ms"].values()) - r["ms"]) < 0.2 * r["ms"] + 0.05
    r12 = gk.bench_msm_g1(logn, c=12, warmup=0, iters=1)
    assert r12["c"] == 12 and r12["result"].tolist() == r["result"].tolist()


def test_cpp_abi_msm_harness(gk, tmp_path):
    """tests/cpp/test_abi_msm.cpp: the MSM entry points from compiled code with plain arrays (the cgo shim's shape), one handle
    shared by four host threads and four handles of their own at once, errors by code -- against the oracle library."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    exe = str(tmp_path / "test_abi_msm")
    lib, orc = os.path.join(ROOT, "gkr-mimc_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_abi_msm.cpp"),
                           "-L" + lib, "-lgkrhip", "-L" + orc, "-lgkr_oracle", "-Wl,-rpath," + lib, "-Wl,-rpath," + orc,
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-L/opt/rocm/lib", "-fopenmp", "-pthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "abi-msm fails=0" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("n,card", [(3, 4), (100, 0), (1000, 2048), (4096, 0), (1 << 14, 0)])
def test_compute_h_then_multi_exp_on_the_device(gk, n, card):
    """h = computeH(a, b, c) then krs2.MultiExp(pk.G1.Z, h) (prove.go:128,221) with H never leaving the device: the same
    point as computeH to the host followed by the MSM on host scalars, and as the oracle's MSM over the returned H."""
    rng = random.Random(7 * n + card)
    a = c.from_ints([rng.randrange(Q) for _ in range(n)])
    b = c.from_ints([rng.randrange(Q) for _ in range(n)])
    cc = c.from_ints([rng.randrange(Q) for _ in range(n)])
    cardinality = card or max(2, 1 << (n - 1).bit_length())
    bases = gk.G1Bases(points=rand_points(n + 3, cardinality))
    got, h = bases.compute_h_multi_exp(a, b, cc, card, want_h=True)
    h_ref = gk.compute_h(a, b, cc, cardinality)
    assert np.array_equal(h, h_ref)
    assert got.tolist() == bases.multi_exp(h_ref).tolist()
    if cardinality <= 4096:
        assert got.tolist() == c.g1_msm(bases.read(), h_ref).tolist()
    assert bases.compute_h_multi_exp(a, b, cc, card).tolist() == got.tolist()
    bases.precompute(0)                 # pk.G1.Z on fixed-base tables: H's limb planes go straight into the tables' digit kernel
    assert bases.compute_h_multi_exp(a, b, cc, card).tolist() == got.tolist()
    bases.precompute(13)
    got2, h2 = bases.compute_h_multi_exp(a, b, cc, card, want_h=True)
    assert got2.tolist() == got.tolist() and np.array_equal(h2, h_ref)
    bases.close()


# ---- the two-level sort (host_msm.hip.h: from 2^20 points by default; forced here at sizes the oracle reaches) ----------
@pytest.fixture
def two_level(gk):
    gk.set_option("msm_sort_levels", 2)
    yield
    gk.set_option("msm_sort_levels", 0)


@pytest.mark.parametrize("n", [1, 2, 5, 100, 1000, 4096, 1 << 14])
def test_msm_two_level_sort_vs_oracle(gk, two_level, n):
    rng = random.Random(300 + n)
    pts = rand_points(n + 2, n)
    sc = rand_scalars(rng, n)
    if n >= 100:
        pts[3] = 0
        pts[5] = pts[4]
        sc[5] = sc[4]
    want = c.g1_msm(pts, sc)
    b = gk.G1Bases(points=pts)
    assert b.multi_exp(sc).tolist() == want.tolist()
    assert b.multi_exp(sc).tolist() == want.tolist()
    assert b.multi_exp(sc[: n // 2]).tolist() == c.g1_msm(pts[: n // 2], sc[: n // 2]).tolist()
    b.close()


@pytest.mark.parametrize("cw", [3, 4, 5, 7, 8, 11, 13, 14, 15, 16])
def test_msm_two_level_sort_every_split(gk, two_level, cw):
    """Coarse bins of 2^lowbits buckets for every split the window sizes give (lowbits = 1 .. 7)."""
    rng = random.Random(cw)
    n = 3000
    pts = rand_points(78, n)
    sc = rand_scalars(rng, n)
    b = gk.G1Bases(points=pts)
    b.set_window(cw)
    assert b.multi_exp(sc).tolist() == c.g1_msm(pts, sc).tolist()
    b.close()


def test_msm_two_level_sort_skew_and_long_bins(gk, two_level):
    """Witness-like scalars at 2^17 points: one coarse bin holds 70 % of a window's entries and is cut into many slices (every
    lane of a wave then names the same counter: the aggregated update), big buckets are listed by the refine pass; window
    sizes with a short top window (a few bins of n / 4 entries); the same vector gives the same point with either sort."""
    n = 1 << 17
    pts = rand_points(17, n)
    rng = random.Random(18)
    sc = ec.scalars_to_image([1 if rng.random() < 0.7 else rng.choice([0, 2, 3, Q - 1, rng.randrange(Q)]) for _ in range(n)])
    want = c.g1_msm(pts, sc)
    full = rand_scalars(rng, n)
    want_full = c.g1_msm(pts, full)
    b = gk.G1Bases(points=pts)
    assert b.multi_exp(sc).tolist() == want.tolist()
    assert b.multi_exp(full).tolist() == want_full.tolist()
    for cw in (14, 11):
        b.set_window(cw)
        assert b.multi_exp(full).tolist() == want_full.tolist(), cw
        assert b.multi_exp(sc).tolist() == want.tolist(), cw
    gk.set_option("msm_sort_levels", 1)
    b.set_window(0)
    assert b.multi_exp(sc).tolist() == want.tolist()
    b.close()


def test_msm_sort_levels_agree_at_2p20(gk):
    """2^20 points (two levels by default): the forced single-level sort gives the same point, and a non-reduced scalar is
    still refused."""
    n = 1 << 20
    rng = np.random.default_rng(77)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 61) - 1)
    b = gk.G1Bases(base=c.G1_GEN, scalars=k)
    r2 = b.multi_exp(s)
    bad = s.copy()
    bad[12345] = np.uint64(0xFFFFFFFFFFFFFFFF)
    with pytest.raises(gk.GkrHipError):
        b.multi_exp(bad)
    try:
        gk.set_option("msm_sort_levels", 1)
        r1 = b.multi_exp(s)
    finally:
        gk.set_option("msm_sort_levels", 0)
    assert c.g1_on_curve(r2) and r1.tolist() == r2.tolist()
    b.close()


def test_msm_sort_levels_agree_at_the_largest_size(gk):
    """2^26 points (the ABI's maximum: 26-bit indices leave five low bits in a coarse entry, 1 024 coarse bins per window, slices
    of 16 384 entries): the two sorts give the same point on the micro-benchmark's device-generated data."""
    res = {}
    try:
        for lv in (1, 2):
            gk.set_option("msm_sort_levels", lv)
            res[lv] = gk.bench_msm_g1(26, warmup=0, iters=1)["result"].tolist()
    finally:
        gk.set_option("msm_sort_levels", 0)
    assert res[1] == res[2] and any(res[1]) and c.g1_on_curve(np.array(res[1], dtype=np.uint64))


def test_page_locked_scalars(gk):
    """gkrhip_host_alloc: the same MSM from page-locked memory (a plain DMA upload) and from pageable memory."""
    n = 5000
    rng = random.Random(41)
    pts = rand_points(41, n)
    sc = rand_scalars(rng, n)
    b = gk.G1Bases(points=pts)
    want = c.g1_msm(pts, sc)
    with gk.PinnedArray(n, 4) as pin:
        assert pin.a.shape == (n, 4) and pin.a.dtype == np.uint64
        pin.a[:] = sc
        assert b.multi_exp(pin.a).tolist() == want.tolist()
        assert b.multi_exp(pin.a[:100]).tolist() == c.g1_msm(pts[:100], sc[:100]).tolist()
    with gk.PinnedArray(0, 4) as empty:
        assert empty.a.shape == (0, 4)
    b.close()


def test_reserved_lanes_serve_a_burst_of_calls(gk):
    """gkrhip_reserve_lanes: lanes created ahead; a burst of concurrent MSMs (each leases one) gives the same points as one
    call after the other."""
    import threading
    gk.reserve_lanes(4)
    with pytest.raises(gk.GkrHipError):
        gk.reserve_lanes(17)
    n = 4000
    rng = random.Random(77)
    pts = rand_points(77, n)
    bs = [gk.G1Bases(points=pts) for _ in range(4)]
    scs = [rand_scalars(rng, n) for _ in range(4)]
    want = [c.g1_msm(pts, s).tolist() for s in scs]
    got = [None] * 4

    def work(i):
        got[i] = bs[i].multi_exp(scs[i]).tolist()

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert got == want
    for b in bs:
        b.close()


# ---------------------------------------------------------------- fixed-base tables (round 6)
@pytest.mark.parametrize("n", [1, 2, 3, 17, 100, 1000, 4096, 1 << 14])
def test_msm_fixed_base_vs_oracle(gk, n):
    """gkrhip_msm_g1_precompute: [2^(c j)] P_i once, then every window of every scalar in ONE bucket space (three levels of the LDS
    counting sort) -- the same affine image as the oracle's per-term double-and-add and as the handle's per-window path, with the
    corner scalars of the recoding, a point at infinity, a repeated point, P and -P in one bucket, and a prefix of the bases."""
    rng = random.Random(600 + n)
    pts = rand_points(n + 11, n)
    if n >= 17:
        pts[3] = 0
        pts[5] = pts[4]
        pts[9] = pts[4]
        pts[11, 4:] = ec.point_to_image(ec.neg(ec.point_from_image(pts[10])))[4:]
        pts[11, :4] = pts[10, :4]
    sc = rand_scalars(rng, n)
    if n >= 17:
        sc[5] = sc[4]
        sc[11] = sc[10]
    want = c.g1_msm(pts, sc)
    b = gk.G1Bases(points=pts)
    assert b.multi_exp(sc).tolist() == want.tolist()                   # per-window path
    for cw in (0, 8, 13, 20):
        b.precompute(cw)
        assert b.multi_exp(sc).tolist() == want.tolist(), cw
        assert b.multi_exp(sc).tolist() == want.tolist(), cw           # buffers reused
        m = n // 2
        assert b.multi_exp(sc[:m]).tolist() == c.g1_msm(pts[:m], sc[:m]).tolist(), cw      # a prefix: the tables keep their stride
    b.precompute(-1)                                                    # tables dropped: the per-window path again
    assert b.multi_exp(sc).tolist() == want.tolist()
    b.close()


@pytest.mark.parametrize("cw", [8, 9, 11, 14, 16, 17, 19, 21, 22])
def test_msm_fixed_base_every_window_size(gk, cw):
    """Window sizes on both sides of the old 16-bit limit, with short top windows (255 = 11 * 22 + 13 = 12 * 21 + 3 ...)."""
    rng = random.Random(cw)
    n = 300
    pts = rand_points(78, n)
    sc = rand_scalars(rng, n)
    b = gk.G1Bases(points=pts)
    b.precompute(cw)
    assert b.multi_exp(sc).tolist() == c.g1_msm(pts, sc).tolist()
    vals = [rng.randrange(Q) for _ in range(n)]                        # MultiExpConfig.ScalarsMont on the tables
    assert b.multi_exp(c.from_ints(vals), scalars_mont=True).tolist() == c.g1_msm(pts, ec.scalars_to_image(vals)).tolist()
    b.close()


def test_msm_fixed_base_skew_and_cancellation(gk):
    """The 0/1 wires of a witness (one huge bucket: the segment path over the table array), n copies of one point (doubling
    branch), total cancellation (infinity = (0, 0)), a scalar that is not an fr.Element refused."""
    n = 6000
    pts = rand_points(6, n)
    rng = random.Random(10)
    b = gk.G1Bases(points=pts)
    b.precompute(16)
    ones = ec.scalars_to_image([1] * n)
    assert b.multi_exp(ones).tolist() == c.g1_msm(pts, ones).tolist()
    sc = ec.scalars_to_image([rng.choice([0, 1, 1, 1, 2, Q - 1]) for _ in range(n)])
    assert b.multi_exp(sc).tolist() == c.g1_msm(pts, sc).tolist()
    bad = sc.copy()
    bad[7] = [0xffffffffffffffff] * 4
    with pytest.raises(Exception):
        b.multi_exp(bad)
    assert b.multi_exp(sc).tolist() == c.g1_msm(pts, sc).tolist()     # the handle is usable after the refusal
    b.close()
    same = np.repeat(pts[:1], n, axis=0)
    s = rng.randrange(Q)
    bs = gk.G1Bases(points=same)
    bs.precompute(12)
    assert bs.multi_exp(ec.scalars_to_image([s] * n)).tolist() == c.g1_scalar_mul(pts[0], ec.scalar_to_limbs(n * s % Q)).tolist()
    bs.close()
    pm = np.concatenate([pts[:100], pts[:100]])
    bp = gk.G1Bases(points=pm)
    bp.precompute(0)
    assert bp.multi_exp(ec.scalars_to_image([5] * 100 + [Q - 5] * 100)).tolist() == [0] * 8
    bp.close()


def test_msm_fixed_base_agrees_with_the_per_window_path_at_2p20(gk):
    """2^20 synthetic bases (beyond the oracle's reach): the fixed-base sum at c = 20 and c = 22 equals the per-window sum, and
    MSM(s) + MSM(t) = MSM(s + t) holds on the tables."""
    n = 1 << 20
    rng = np.random.default_rng(20)
    k = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 59) - 1)
    b = gk.G1Bases(base=c.G1_GEN, scalars=k)
    s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 59) - 1)
    want = b.multi_exp(s).tolist()
    for cw in (20, 22):
        b.precompute(cw)
        assert b.multi_exp(s).tolist() == want, cw
    # the 0/1 wires of a witness on the tables: one bucket holds 40 % of all entries of window 0 (segments, slices of one bin)
    w = s.copy()
    w[: n // 5 * 2] = 0
    w[: n // 5 * 2, 0] = 1
    w[n // 5 * 2: n // 5 * 4] = 0
    b.precompute(22)
    got = b.multi_exp(w).tolist()
    b.precompute(-1)
    assert got == b.multi_exp(w).tolist()
    b.close()


def test_bench_msm_fixed_base_result(gk):
    logn = 9
    n = 1 << logn
    r = gk.bench_msm_g1_fixed_base(logn, warmup=1, iters=2)
    ks = _synth_scalars(n, 0x1234567)
    ss = _synth_scalars(n, 0x7654321)
    tot = sum(a * b for a, b in zip(ks, ss)) % Q
    assert r["result"].tolist() == c.g1_scalar_mul(c.G1_GEN, ec.scalar_to_limbs(tot)).tolist()
    assert r["ms"] > 0 and r["precompute_ms"] > 0 and r["c"] >= 8
    r13 = gk.bench_msm_g1_fixed_base(logn, c=13, warmup=0, iters=1)
    assert r13["c"] == 13 and r13["result"].tolist() == r["result"].tolist()
