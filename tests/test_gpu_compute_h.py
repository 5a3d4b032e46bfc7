"""GPU parity tests of computeH (prover/gadget/prove.go:308-359) through the C ABI: bit-exact against
the Python restatement of gnark-crypto's fft.Domain (oracle/pyoracle_fft.py -- un-vendored dependency, "parity unpinned":
pinned on the published algorithm and on schoolbook division, tests/test_oracle.py), against the committed fixtures, and --
at sizes no big-int oracle reaches in seconds -- through the identity H * (X^n - 1) = A*B - C at a random point."""
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
import pyoracle_fft as F  # noqa: E402
from util import load  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gk():
    g = importlib.import_module("gkr-mimc_amd")
    g.init(0)
    return g


def words(vals):      # regular-form integers -> (n, 4) uint64 words
    return np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in vals], dtype=np.uint64).reshape(-1, 4)


def ints(arr):
    return [sum(int(row[k]) << (64 * k) for k in range(4)) for row in np.asarray(arr).reshape(-1, 4)]


def test_compute_h_golden(gk):
    for e in load("compute_h.json"):
        a, b, cc = ([int(x, 16) for x in e[k]] for k in ("a", "b", "c"))
        h = gk.compute_h(c.from_ints(a), c.from_ints(b), c.from_ints(cc), e["cardinality"])
        assert [hex(v) for v in ints(h)] == e["h"], (e["n"], e["cardinality"])


@pytest.mark.parametrize("n,card", [(1, 0), (2, 0), (4, 0), (7, 8), (8, 0), (9, 0), (32, 0), (100, 0), (100, 512), (1024, 0),
                                    (3000, 0), (4096, 0), (5000, 16384)])
def test_compute_h_vs_oracle(gk, n, card):
    """Every pass shape (one, two, three stages per pass; a transform of one to five passes), padding, explicit domains;
    satisfied constraints on part of the vector, arbitrary values (and the corner values 0, 1, q - 1) elsewhere."""
    rng = random.Random(1000 * n + card)
    a = [rng.randrange(F.Q) for _ in range(n)]
    b = [rng.randrange(F.Q) for _ in range(n)]
    cc = [(x * y) % F.Q if i % 3 else rng.randrange(F.Q) for i, (x, y) in enumerate(zip(a, b))]
    for i, v in enumerate((0, 1, F.Q - 1)):
        if i < n:
            a[i] = v
            cc[-1 - i] = v
    want = F.compute_h(a, b, cc, card or None)
    got = gk.compute_h(c.from_ints(a), c.from_ints(b), c.from_ints(cc), card)
    assert got.shape[0] == len(want)
    assert ints(got) == want


def test_compute_h_refuses_bad_arguments(gk):
    a = c.from_ints([1, 2, 3])
    with pytest.raises(gk.GkrHipError, match="not a power of two"):
        gk.compute_h(a, a, a, 6)
    with pytest.raises(gk.GkrHipError, match="not a power of two"):
        gk.compute_h(a, a, a, 2)                      # smaller than the vectors
    bad = a.copy()
    bad[1] = np.array([0xFFFFFFFFFFFFFFFF] * 4, dtype=np.uint64)
    with pytest.raises(gk.GkrHipError, match="canonical"):
        gk.compute_h(bad, a, a)


def _eval_interpolant(vals, dom, tau):
    """P(tau) for the polynomial of degree < n with P(g^i) = vals[i]: barycentric formula, one batch inversion."""
    n = dom.cardinality
    xs, g = [], 1
    for _ in range(n):
        xs.append((tau - g) % F.Q)
        g = g * dom.generator % F.Q
    pref = [1] * (n + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * x % F.Q
    inv = pow(pref[n], F.Q - 2, F.Q)
    acc, g = 0, 1
    gs = [1] * n
    for i in range(1, n):
        gs[i] = gs[i - 1] * dom.generator % F.Q
    for i in range(n - 1, -1, -1):
        xi_inv = inv * pref[i] % F.Q
        inv = inv * xs[i] % F.Q
        acc = (acc + vals[i] * gs[i] % F.Q * xi_inv) % F.Q
    return (pow(tau, n, F.Q) - 1) * dom.cardinality_inv % F.Q * acc % F.Q


@pytest.mark.parametrize("logn", [16, 19])
def test_compute_h_quotient_identity_at_full_size(gk, logn):
    """Sizes beyond the big-int oracle: with c = a*b on the domain, H * (tau^n - 1) == A(tau) * B(tau) - C(tau) at a random
    tau (A, B, C evaluated from the inputs by the barycentric formula, H from the returned coefficients)."""
    n = 1 << logn
    dom = F.Domain(n, 1)
    a = c.random_fr_array(n)
    b = c.from_ints([(3 * i * i + 7 * i + 11) % 1000000007 for i in range(n)])
    cc = c.fr(n)
    for i in range(n):                                    # c = a * b pointwise (oracle field multiplication)
        c.lib.oracle_fr_mul(cc[i:].ctypes.data, a[i:].ctypes.data, b[i:].ctypes.data)
    h = ints(gk.compute_h(a, b, cc))
    tau = random.Random(logn).randrange(F.Q)
    A, B, C = (_eval_interpolant(c.to_ints(v), dom, tau) for v in (a, b, cc))
    Htau, shift = 0, dom.log
    pw = [1] * n
    for k in range(1, n):
        pw[k] = pw[k - 1] * tau % F.Q
    for p in range(n):                                    # position p holds coefficient rev(p)
        Htau = (Htau + h[p] * pw[F.bit_reverse(p, shift)]) % F.Q
    assert Htau * (pow(tau, n, F.Q) - 1) % F.Q == (A * B - C) % F.Q
