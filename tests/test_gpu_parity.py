"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libgkrhip.so via gkr-mimc_amd/prover.py), against the CPU oracle on identical inputs -- bit-exact,
as all of this is integer field arithmetic -- against the committed golden fixtures, and, at the
BASELINE sizes the oracle cannot reach in seconds, through size-independent properties (verifier
acceptance, claims == Evaluate(layer, point), repeatability).

Reads like the reference's own tests: poly/multilin_test.go, poly/eq_test.go,
circuit/gates/gates_test.go, sumcheck/prover_test.go, gkr/gkr_test.go."""
import hashlib
import importlib

import numpy as np
import pytest

import coracle as c
import pyoracle as o
from util import fr_to_hex, hex_to_fr, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gk():
    mod = importlib.import_module("gkr-mimc_amd")
    mod.init(0)  # fails loudly without a gfx950 GPU / built library: no fallback
    return mod


def nasty(n, seed=1):
    """Canonical elements whose 32-bit limbs are frequently 0xFFFFFFFF / 0 (carry corner cases)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    pat = rng.integers(0, 4, size=(n, 4))
    a[pat == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    a[pat == 1] = np.uint64(0xFFFFFFFF00000000)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)  # < 2^252 < q : canonical
    a[0] = [0, 0, 0, 0]
    if n > 1:
        a[1] = [0x43e1f593f0000000, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]  # q - 1
    return a


# ---------------------------------------------------------------- field arithmetic through kernels
def test_cipher_gate_eval_batch_nasty_limbs(gk):
    # circuit/gates/gates_test.go:31-40 (EvalBatch == Eval) with carry-corner inputs
    n = 1 << 12
    L, R = nasty(n, 1), nasty(n, 2)
    ark = c.from_u64(25)
    got = gk.gate_eval_batch(gk.GATE_CIPHER, ark, [L, R])
    want = c.fr(n)
    c.lib.oracle_gate_eval_batch(c.GATE_CIPHER, ark.ctypes.data, want.ctypes.data, c._ptr_array([L, R]), 2, n)
    assert np.array_equal(got, want)
    got = gk.gate_eval_batch(gk.GATE_IDENTITY, None, [L, R])
    assert np.array_equal(got, L)


# ---------------------------------------------------------------- poly.MultiLin.Fold
def test_fold_kat(gk):
    # poly/multilin_test.go:12-31
    out = gk.fold(c.from_ints([0, 1, 2, 3]), c.from_u64(5))
    assert c.to_ints(out) == [10, 11]


@pytest.mark.parametrize("bn", [1, 2, 3, 7, 10, 16, 20])
def test_fold_vs_oracle(gk, bn):
    t = c.random_fr_array(1 << bn) if bn % 2 else nasty(1 << bn, bn)
    r = c.mimc_hash(c.from_u64(bn))
    assert np.array_equal(gk.fold(t, r), c.fold(t, r))


def test_fold_golden(gk):
    for e in load("poly.json")["fold"]:
        assert fr_to_hex(gk.fold(hex_to_fr(e["tbl"]), hex_to_fr(e["r"]))) == e["out"]


def test_fold_rejects_bad_length(gk):
    with pytest.raises(gk.prover.GkrHipError):
        gk.fold(c.random_fr_array(6), c.from_u64(5))
    with pytest.raises(gk.prover.GkrHipError):
        gk.fold(c.random_fr_array(1), c.from_u64(5))


def test_fold_linearity_full_size(gk):
    """2^24 elements (BASELINE config 3 table size): fold(a,r) + fold(b,r) == fold(a+b,r) on a sample,
    and folding twice equals Evaluate-style recursion on the oracle for a slice."""
    n = 1 << 24
    a = c.random_fr_array(n)
    r = c.mimc_hash(c.from_u64(77))
    fa = gk.fold(a, r)
    # spot-check 4096 positions against the oracle's per-element formula
    idx = np.random.default_rng(0).integers(0, n // 2, 4096)
    sub = np.concatenate([a[idx], a[idx + n // 2]])  # pairs (i, i+mid) re-packed as a small table
    assert np.array_equal(fa[idx], c.fold(sub, r))


# ---------------------------------------------------------------- poly.FoldedEqTable / EvalEq
@pytest.mark.parametrize("bn", list(range(0, 15)) + [17])
def test_eq_table_vs_oracle(gk, bn):
    q = c.random_fr_array(bn)
    assert np.array_equal(gk.folded_eq_table(q), c.folded_eq_table(q))
    m = c.mimc_hash(c.from_u64(bn + 1))
    assert np.array_equal(gk.folded_eq_table(q, m), c.folded_eq_table(q, m))


@pytest.mark.parametrize("bn", range(0, 13))
def test_eq_table_evaluate_equals_eval_eq(gk, bn):
    # poly/eq_test.go:12-26: EvalEq(q,h) == FoldedEqTable(q).Evaluate(h), both on the GPU path
    q, h = c.random_fr_array(bn), c.mimc_hash(c.from_u64(3)).repeat(bn, axis=0) if bn else c.fr(0)
    eq = gk.folded_eq_table(q)
    assert np.array_equal(gk.evaluate(eq, h), c.eval_eq(q, h) if bn else c.from_u64(1))


@pytest.mark.parametrize("bn,chunk", [(1, 1), (4, 2), (9, 256), (12, 256), (12, 4096), (13, 64)])
def test_chunk_of_eq_table_vs_oracle(gk, bn, chunk):
    """poly.ChunkOfEqTable (poly/eq.go:61-89): every chunk lands at its place and the chunks together are the whole
    table (the reference asserts the same, poly/eq_test.go:28-58), with and without a multiplier."""
    q = c.random_fr_array(bn)
    mult = c.mimc_hash(c.from_u64(bn))
    for m in (None, mult):
        want = c.folded_eq_table(q, m)
        assert np.array_equal(c.chunked_eq_table(q, chunk, m), want)
        table = np.zeros((1 << bn, 4), np.uint64)
        n_chunks = (1 << bn) // chunk
        ids = range(n_chunks) if n_chunks <= 8 else [0, 1, n_chunks // 2, n_chunks - 1]
        for cid in ids:
            gk.chunk_of_eq_table(table, cid, chunk, q, m)
            assert np.array_equal(table[cid * chunk:(cid + 1) * chunk], want[cid * chunk:(cid + 1) * chunk]), (bn, chunk, cid)
        if n_chunks <= 8:
            assert np.array_equal(table, want)


def test_eq_golden(gk):
    for e in load("poly.json")["eq"]:
        m = hex_to_fr(e["mult"]) if e["mult"] else None
        assert fr_to_hex(gk.folded_eq_table(hex_to_fr(e["q"]).reshape(-1, 4), m)) == e["out"]


@pytest.mark.parametrize("bn", [0, 1, 5, 12, 18])
def test_evaluate_vs_oracle(gk, bn):
    t = c.random_fr_array(1 << bn)
    pt = nasty(bn + 2, 5)[2:]
    assert np.array_equal(gk.evaluate(t, pt), c.evaluate(t, pt))


# ---------------------------------------------------------------- sumcheck.Prove
def _cipher_instance(bn):
    n = 1 << bn
    X = [c.from_ints(range(n)), c.from_ints(range(n))]
    ark = c.from_u64(145646)
    qs = c.random_fr_array(bn).reshape(1, bn, 4)
    claims = c.evaluation(c.GATE_CIPHER, ark, qs, c.fr(0), X)
    return X, claims, qs, ark


def _multi_instance(bn, ninst):
    n = 1 << bn
    X = [c.from_ints(range(n)), c.from_ints(range(n))]
    qs = np.stack([c.from_ints([(i * j + i) for j in range(bn)]).reshape(bn, 4) for i in range(ninst)])
    claims = np.concatenate([c.evaluation(c.GATE_IDENTITY, None, qs[i:i + 1], c.fr(0), X) for i in range(ninst)])
    return X, claims, qs


def test_sumcheck_golden(gk):
    for e in load("sumcheck.json"):
        bn = e["bn"]
        n = 1 << bn
        X = [c.from_ints(range(n)), c.from_ints(range(n))]
        qs = np.stack([hex_to_fr(q).reshape(bn, 4) for q in e["qprimes"]]) if bn else np.zeros((len(e["qprimes"]), 0, 4), np.uint64)
        gate = gk.GATE_CIPHER if e["kind"] == "cipher" else gk.GATE_IDENTITY
        ark = hex_to_fr(e["ark"]) if e["ark"] else None
        proof, chal, final = gk.sumcheck_prove(X, qs, hex_to_fr(e["claims"]), gate, ark)
        assert [fr_to_hex(r) for r in proof] == e["proof"]
        assert fr_to_hex(chal) == e["challenges"]
        assert fr_to_hex(final) == e["final"]


@pytest.mark.parametrize("bn", range(0, 15))
def test_sumcheck_cipher_vs_oracle(gk, bn):
    # sumcheck/prover_test.go:88-94 (TestWithCipherGate) + genericTest's verifier checks
    X, claims, qs, ark = _cipher_instance(bn)
    proof, chal, final = gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
    oproof, ochal, ofinal = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims)
    assert np.array_equal(proof, oproof) and np.array_equal(chal, ochal) and np.array_equal(final, ofinal)
    rc, vchal, expected, _ = c.sumcheck_verify(claims, proof)
    assert rc == 0 and np.array_equal(vchal, chal)
    fin = c.to_ints(final)
    assert o.CipherGate(145646).eval(*fin[1:]) * fin[0] % o.Q == c.to_ints(expected)[0]


@pytest.mark.parametrize("bn", range(0, 15))
def test_sumcheck_multi_identity_vs_oracle(gk, bn):
    # sumcheck/prover_test.go:80-86 (TestWithMultiIdentity, 10 instances)
    X, claims, qs = _multi_instance(bn, 10)
    proof, chal, final = gk.sumcheck_prove(X, qs, claims, gk.GATE_IDENTITY)
    oproof, ochal, ofinal = c.sumcheck_prove(c.GATE_IDENTITY, None, X, qs, claims)
    assert np.array_equal(proof, oproof) and np.array_equal(chal, ochal) and np.array_equal(final, ofinal)
    rc, vchal, _, recomb = c.sumcheck_verify(claims, proof)
    assert rc == 0 and np.array_equal(vchal, chal)
    assert np.array_equal(recomb, c.mimc_hash(claims))


@pytest.mark.parametrize("bn", [1, 2, 3, 6, 9, 10, 13, 17, 18])
def test_sumcheck_linear_gates_single_point_vs_oracle(gk, bn):
    """One evaluation point and a linear gate (identity with one and with two tables, the add gate): the fused
    linear rounds (linear_round.hip.h) against the oracle; carry-corner tables; the claim is false on purpose for
    the add gate (the transcript does not depend on it)."""
    n = 1 << bn
    q = c.random_fr_array(bn).reshape(1, bn, 4)
    A, B = nasty(n, bn + 40), c.random_fr_array(n)
    ark = c.from_u64(3 * bn + 1)
    for gate, X, a, cl in ((gk.GATE_IDENTITY, [A], None, c.evaluation(c.GATE_IDENTITY, None, q, c.fr(0), [A])),
                           (gk.GATE_IDENTITY, [B, A], None, c.evaluation(c.GATE_IDENTITY, None, q, c.fr(0), [B, A])),
                           (gk.GATE_ADD, [A, B], ark, c.from_u64(77))):
        proof, chal, final = gk.sumcheck_prove(X, q, cl, gate, a)
        oproof, ochal, ofinal = c.sumcheck_prove(gate, a, X, q, cl)
        assert np.array_equal(proof, oproof) and np.array_equal(chal, ochal) and np.array_equal(final, ofinal), gate


@pytest.mark.parametrize("bn", [0, 1, 2, 5, 9, 14])
def test_sumcheck_generic_test_like_the_reference(gk, bn):
    """sumcheck/prover_test.go:42-94 (genericTest over InitializeCipherGateInstance / InitializeMultiInstance(bn, 10)):
    the claims recombine to Evaluation, the restated sumcheck.Verify accepts the GPU prover's messages with the SAME
    challenges, and gate(finalClaims[1:]) * finalClaims[0] is the verifier's expected value."""
    n = 1 << bn
    tab = c.from_ints(list(range(n)))
    for kind in ("cipher", "multi"):
        if kind == "cipher":
            gate, ogate, ark = gk.GATE_CIPHER, c.GATE_CIPHER, c.from_u64(145646)
            qs = c.random_fr_array(bn).reshape(1, bn, 4)
        else:
            gate, ogate, ark = gk.GATE_IDENTITY, c.GATE_IDENTITY, None
            qs = np.stack([c.from_ints([(i * j + i) % o.Q for j in range(bn)]).reshape(bn, 4) for i in range(10)])
        X = [tab, tab.copy()]
        claims = np.concatenate([c.evaluation(ogate, ark, qs[i:i + 1], c.fr(0), X) for i in range(qs.shape[0])])
        proof, chal, fin = gk.sumcheck_prove(X, qs, claims, gate, ark)
        rc, vchal, expected, recomb = c.sumcheck_verify(claims, proof)
        assert rc == 0 and np.array_equal(vchal, chal)
        # the verifier's final check (gkr/verifier.go:93-110 does the same with the gate)
        gate_val = c.gate_eval_batch(ogate, ark, [fin[1:2], fin[2:3]])
        prod = c.fr()
        c.lib.oracle_fr_mul(c._p(prod), c._p(np.ascontiguousarray(gate_val)), c._p(np.ascontiguousarray(fin[0:1])))
        assert np.array_equal(prod, expected), (kind, bn)


def test_sumcheck_91_claims(gk):
    # shape of MiMC layer 2 / BenchmarkMultiIdentity (sumcheck/prover_test.go:111-125) at bn = 10
    X, claims, qs = _multi_instance(10, 91)
    proof, chal, final = gk.sumcheck_prove(X[:1], qs, claims, gk.GATE_IDENTITY)
    oproof, ochal, ofinal = c.sumcheck_prove(c.GATE_IDENTITY, None, X[:1], qs, claims)
    assert np.array_equal(proof, oproof) and np.array_equal(chal, ochal) and np.array_equal(final, ofinal)


def test_sumcheck_nasty_tables(gk):
    bn = 9
    X = [nasty(1 << bn, 11), nasty(1 << bn, 12)]
    ark = nasty(3, 13)[2:3]
    qs = nasty(bn + 2, 14)[2:].reshape(1, bn, 4)
    claims = c.evaluation(c.GATE_CIPHER, ark, qs, c.fr(0), X)
    got = gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
    want = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_sumcheck_does_not_mutate_inputs_and_checks_sizes(gk):
    X, claims, qs, ark = _cipher_instance(6)
    keep = [x.copy() for x in X]
    gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
    assert all(np.array_equal(a, b) for a, b in zip(X, keep))
    with pytest.raises(gk.prover.GkrHipError):  # sumcheck/prover.go:52-56
        gk.sumcheck_prove([X[0], X[1][:32]], qs, claims, gk.GATE_CIPHER, ark)
    Xm, claims_m, qs_m = _multi_instance(4, 3)
    with pytest.raises(gk.prover.GkrHipError):  # sumcheck/prover.go:113-115
        gk.sumcheck_prove(Xm, qs_m, claims_m[:2], gk.GATE_IDENTITY)


def test_sumcheck_output_does_not_depend_on_claims_being_true(gk):
    """In the reference the claims only feed Fiat-Shamir (sumcheck/prover.go:36-37,128): a caller may pass any
    values.  sumcheck.Prove through the C ABI therefore never derives a round sum from the claim (that shortcut
    is internal to gkr.Prove): with false claims the output still equals the oracle's, and for a single point it
    equals the output with the true claim."""
    X, claims, qs, ark = _cipher_instance(8)
    bogus = c.mimc_hash(claims)
    got = gk.sumcheck_prove(X, qs, bogus, gk.GATE_CIPHER, ark)
    want = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, bogus)
    true = gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
    for a, b, t in zip(got, want, true):
        assert np.array_equal(a, b) and np.array_equal(a, t)
    Xm, claims_m, qs_m = _multi_instance(7, 5)
    bogus_m = np.concatenate([c.mimc_hash(claims_m[i:i + 1]) for i in range(5)])
    got = gk.sumcheck_prove(Xm, qs_m, bogus_m, gk.GATE_IDENTITY)
    want = c.sumcheck_prove(c.GATE_IDENTITY, None, Xm, qs_m, bogus_m)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # degenerate evaluation points (coordinates 0 and 1) exercise the eq-weight corner cases
    qz = c.from_ints([0, 1, 0, 1, 1, 0, 5, 0]).reshape(1, 8, 4)
    cl = c.evaluation(c.GATE_CIPHER, ark, qz, c.fr(0), X)
    got = gk.sumcheck_prove(X, qz, cl, gk.GATE_CIPHER, ark)
    want = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qz, cl)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_sumcheck_cipher_bn20_vs_oracle(gk):
    """BASELINE config 2 size (bN = 20) on the reference's benchmark instance shape
    (sumcheck/prover_test.go:96-109: L[i]=R[i]=i, ark=145646)."""
    bn = 20
    n = 1 << bn
    L = c.fr(n)
    c.lib.oracle_random_fr_array(L.ctypes.data, n)  # any table works; cheap to build
    X = [L, L.copy()]
    ark = c.from_u64(145646)
    qs = c.random_fr_array(bn).reshape(1, bn, 4)
    claims = c.evaluation(c.GATE_CIPHER, ark, qs, c.fr(0), X)
    proof, chal, final = gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
    oproof, ochal, ofinal = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims)
    assert np.array_equal(proof, oproof) and np.array_equal(chal, ochal) and np.array_equal(final, ofinal)


def _run_case(env, sizes, circuit="mimc"):
    import os, subprocess, sys
    e = dict(os.environ)
    e.update(env)
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "gpu_case.py"), sizes, circuit], env=e, capture_output=True,
                         text=True, timeout=1200)
    assert out.returncode == 0 and "CASE-OK" in out.stdout, out.stdout + out.stderr


def test_generic_partial_eval_path(gk):
    """The reference-shaped evaluator (t = 0..8, Eq table folded) stays available and bit-identical:
    GKRHIP_GENERIC=1 routes cipher layers through k_partial_eval + k_fold instead of k_cipher_round."""
    _run_case({"GKRHIP_GENERIC": "1"}, "1,2,5,9,12")
    _run_case({"GKRHIP_GMAX": "8"}, "9,11", circuit="gmimc")       # fused linear rounds with several pairs per lane
    _run_case({"GKRHIP_GENERIC": "1"}, "3,8", circuit="gmimc")     # the same circuit through the reference-shaped rounds


def test_round_kernel_small_thread_budget(gk):
    """GKRHIP_GMAX=8 caps the round kernel at 256 threads so the per-iteration eq factor (Wj), the
    multi-iteration loop and the multi-block hand-off are exercised at small sizes."""
    _run_case({"GKRHIP_GMAX": "8"}, "1,3,8,9,10,13")
    _run_case({"GKRHIP_GMAX": "10"}, "11,12,14")


def test_eq_pyramid_in_two_launches(gk):
    """The per-lane eq pyramid built in two launches (levels up to 2^n entries with short chains, the upper levels with one
    product per entry from a small second pyramid) and in one: same weights, same transcript, for splits below, at and above
    the pyramid's height, with shard-like thread budgets."""
    _run_case({"GKRHIP_GMAX": "8", "GKR_CASE_OPTIONS": "pyr_split=3"}, "9,11,13")
    _run_case({"GKR_CASE_OPTIONS": "pyr_split=1"}, "5,10,14")
    _run_case({"GKR_CASE_OPTIONS": "pyr_split=0"}, "9,14")
    _run_case({"GKR_CASE_OPTIONS": "pyr_split=12"}, "14,15,16")
    _run_case({"GKRHIP_GMAX": "10", "GKR_CASE_OPTIONS": "pyr_split=5"}, "12", circuit="gmimc")


def test_round_kernel_variants(gk):
    """The throughput kernel everywhere (GKRHIP_LAT=0) and the interleaved-pair kernel everywhere
    (GKRHIP_LAT=2, with and without the per-iteration eq factor) give the same transcript."""
    _run_case({"GKRHIP_LAT": "0"}, "1,4,9,12")
    _run_case({"GKR_CASE_OPTIONS": "claim_trick=0"}, "1,2,7,12")   # all eight monomial sums computed on the device
    _run_case({"GKRHIP_LAT": "2", "GKRHIP_GMAX": "8"}, "1,4,9,12,13")


def test_host_tail_rounds(gk):
    """GKRHIP_HOST_TAIL = h: the device exports the tables of the round with 2^(h+1) pairs and the host runs the last
    h+1 rounds of every single-point cipher sumcheck itself -- the same transcript for every h, also with the
    throughput kernel as the exporting round and when the sumcheck is too short to have a device round at all."""
    for h in ("0", "1", "3", "5", "6"):     # 0: every round on the device (the default is 4)
        _run_case({"GKRHIP_HOST_TAIL": h}, "1,2,3,5,8,9,12")
    _run_case({"GKRHIP_HOST_TAIL": "0"}, "2,9", circuit="gmimc")
    _run_case({"GKRHIP_HOST_TAIL": "3"}, "3,9,11", circuit="gmimc")
    _run_case({"GKRHIP_HOST_TAIL": "5", "GKRHIP_LAT": "0"}, "7,8,11")
    # h = 7..10 (kHostTailMax was raised from 6 to 10 in round 5: 2^8..2^11 pairs per table through h_tail, the speculative export
    # and, for t > 6, without round 0 ahead of its point): against the oracle, not only a bench sweep with the native verifier
    _run_case({"GKRHIP_HOST_TAIL": "8", "GKRHIP_LAT": "0"}, "9,10,12,14")
    _run_case({"GKRHIP_HOST_TAIL": "10", "GKRHIP_LAT": "0"}, "11,12,13,15")
    _run_case({"GKRHIP_HOST_TAIL": "8", "GKRHIP_LAT": "0"}, "10,12", circuit="gmimc")
    _run_case({"GKRHIP_HOST_TAIL": "10", "GKRHIP_LAT": "0"}, "12,13", circuit="gmimc")
    _run_case({"GKRHIP_HOST_TAIL": "2", "GKRHIP_GMAX": "8", "GKR_CASE_OPTIONS": "claim_trick=0"}, "4,10,13")


def test_round_kernel_deferred_reduction_variants(gk):
    """The deferred-reduction round kernel (wide LDS/VGPR accumulators, one reduction per lane and sum) against the
    oracle with the lane weight applied after the loop from 2 pairs per lane on, never, and with the kernel
    switched off (every product reduced) -- the same transcript each time."""
    _run_case({"GKRHIP_GMAX": "8", "GKR_CASE_OPTIONS": "wt_late_lj=1"}, "9,10,12,14")
    _run_case({"GKRHIP_GMAX": "8", "GKR_CASE_OPTIONS": "wt_late_lj=99"}, "9,10,13")
    _run_case({"GKRHIP_GMAX": "9", "GKRHIP_WIDE": "0"}, "10,12,13")
    _run_case({"GKRHIP_GMAX": "8", "GKRHIP_LAT": "0", "GKR_CASE_OPTIONS": "wt_late_lj=3"}, "11,15")
    # a proof alone on the GPU doubles the threads of its big rounds (two eq-weight splits in one sumcheck); off:
    _run_case({"GKRHIP_GMAX": "8", "GKR_CASE_OPTIONS": "solo_boost=0"}, "10,11,12")


def test_prelaunched_rounds(gk):
    """Round k+1's kernel queued before round k is hashed (it polls the host-mapped challenge slot): the same transcript
    with the pre-launch forced on for every round size, off, and combined with the other round variants; the counter
    proves the path ran."""
    on = {"GKRHIP_PRELAUNCH": "2", "GKRHIP_CASE_EXPECT": "prelaunched_rounds"}
    _run_case(on, "1,2,3,5,9,12,14")
    _run_case(dict(on, GKRHIP_HOST_TAIL="0"), "2,6,10")
    _run_case(dict(on, GKRHIP_GMAX="8", GKRHIP_HOST_TAIL="3"), "9,11,13")
    _run_case(dict(on, GKRHIP_GMAX="8", GKRHIP_LAT="0", GKR_CASE_OPTIONS="claim_trick=0"), "8,12")
    _run_case(dict(on, GKRHIP_GMAX="8"), "3,9,11", circuit="gmimc")           # the linear rounds too
    _run_case({"GKRHIP_PRELAUNCH": "0", "GKRHIP_CASE_EXPECT_NOT": "prelaunched_rounds"}, "2,9,12")
    _run_case({"GKRHIP_PRELAUNCH": "1", "GKRHIP_HOST_TAIL": "0", "GKRHIP_CASE_EXPECT": "prelaunched_rounds", "GKR_CASE_OPTIONS": "prelaunch_lg=4"}, "7,10")


def test_cooperative_small_rounds(gk):
    """k_cipher_round_coop (eight lanes per pair, products dealt level by level through LDS) for the small rounds: same
    transcript with the kernel forced on for every size it takes, with the last round and the host-tail export inside
    it, several iterations per workgroup, all eight sums on the device, pre-launched or not, and switched off."""
    on = {"GKRHIP_COOP": "2", "GKRHIP_SPEC": "0", "GKRHIP_CASE_EXPECT": "coop_rounds"}   # (the speculative rounds would take the small ones)
    _run_case(on, "1,2,3,5,6,9,12,14")
    _run_case(dict(on, GKRHIP_HOST_TAIL="0"), "1,2,4,7,10,13")               # the P = 1 round and its tail words
    _run_case(dict(on, GKRHIP_HOST_TAIL="3", GKR_CASE_OPTIONS="coop_wgs=2"), "6,9,11")  # export + several iterations per workgroup
    _run_case(dict(on, GKRHIP_PRELAUNCH="0", GKR_CASE_OPTIONS="claim_trick=0"), "3,8,12")
    _run_case(dict(on, GKRHIP_PRELAUNCH="2", GKRHIP_HOST_TAIL="1"), "5,10,14")
    _run_case(dict(on, GKRHIP_COOP_LG="6", GKRHIP_GMAX="8"), "9,12")
    _run_case(dict(on, GKRHIP_HOST_TAIL="2"), "4,9", circuit="gmimc")
    _run_case({"GKRHIP_COOP": "0", "GKRHIP_CASE_EXPECT_NOT": "coop_rounds"}, "3,9,12")


def test_speculative_small_rounds(gk):
    """k_cipher_round_spec: the small rounds run for the eight candidate values 0..7 of the previous challenge while the host
    is still hashing, and the host interpolates the candidates at the true challenge (degree 7 in r: exact).  Same transcript
    with the path forced on -- every start round the sizes allow (first speculative round = the export round; several
    speculative rounds with the tables alternating between two buffers; start at round 2 straight off the first fold), all
    eight sums on the device, every host-tail depth, beside the cooperative kernel and the look-ahead -- and switched off."""
    on = {"GKRHIP_SPEC": "2", "GKRHIP_CASE_EXPECT": "spec_rounds"}
    _run_case(on, "7,8,9,10,12,14,16")
    _run_case(dict(on, GKRHIP_SPEC_LG="5"), "8,9,11,13")                  # one or two speculative rounds just before the host takes over
    _run_case(dict(on, GKRHIP_SPEC_LG="16", GKRHIP_GMAX="16"), "9,13,15")  # from round 2 on
    _run_case(dict(on, GKRHIP_HOST_TAIL="1"), "5,6,9,12")
    _run_case(dict(on, GKRHIP_HOST_TAIL="6"), "9,10,13")
    _run_case(dict(on, GKR_CASE_OPTIONS="claim_trick=0"), "8,11")                   # M_0 from the candidates as well
    _run_case(dict(on, GKRHIP_COOP="0", GKRHIP_PRE="0"), "9,12")
    _run_case(dict(on, GKRHIP_PRE="2", GKRHIP_GMAX="8", GKRHIP_CASE_EXPECT="spec_rounds,lookahead_round0"), "11,13")
    # the GMiMC circuit: cipher layers and LINEAR layers (add, copy: k_linear_round_spec, two candidates -- their sums are linear in r)
    _run_case(dict(on, GKRHIP_HOST_TAIL="3"), "7,10", circuit="gmimc")
    _run_case(on, "9,12,13", circuit="gmimc")
    _run_case(dict(on, GKRHIP_HOST_TAIL="1", GKRHIP_SPEC_LG="6"), "5,6,8", circuit="gmimc")
    _run_case(dict(on, GKRHIP_HOST_TAIL="2"), "7,11", circuit="gmimc")
    _run_case(dict(on, GKRHIP_HOST_TAIL="4", GKR_CASE_OPTIONS="claim_trick=0"), "8,10", circuit="gmimc")
    _run_case(dict(on, GKRHIP_SPEC="0", GKRHIP_CASE_EXPECT="", GKRHIP_CASE_EXPECT_NOT="spec_rounds"), "9", circuit="gmimc")
    _run_case({"GKRHIP_CASE_EXPECT": "spec_rounds,prelaunched_rounds,coop_rounds"}, "12,15")   # the defaults, alone on the GPU
    _run_case({"GKRHIP_SPEC": "0", "GKRHIP_CASE_EXPECT_NOT": "spec_rounds"}, "9,12")
    _run_case({"GKRHIP_SPEC": "2", "GKRHIP_PRELAUNCH": "0", "GKRHIP_CASE_EXPECT_NOT": "spec_rounds"}, "9,12")    # needs the pre-launched rounds
    _run_case({"GKRHIP_SPEC": "2", "GKRHIP_HOST_TAIL": "0", "GKRHIP_CASE_EXPECT_NOT": "spec_rounds"}, "9,12")    # ... and the host tail


def test_speculative_rounds_random_settings(gk):
    """The speculative path under seeded random combinations of everything that moves its boundaries: the size, where the host
    takes over, the largest round the path takes, the thread limit of a round (which decides where one-pair-per-lane rounds
    begin), polling or argument launches, with and without the cooperative kernel and the look-ahead beside it."""
    import random
    rng = random.Random(20261003)
    for _ in range(10):
        env = {"GKRHIP_SPEC": "2", "GKRHIP_CASE_EXPECT": "spec_rounds",
               "GKRHIP_HOST_TAIL": str(rng.randint(1, 6)), "GKRHIP_SPEC_LG": str(rng.randint(5, 16)),
               "GKRHIP_GMAX": str(rng.choice([8, 10, 12, 16])),
               "GKRHIP_COOP": str(rng.choice([0, 2])), "GKRHIP_PRE": str(rng.choice([0, 2])),
               "GKR_CASE_OPTIONS": "prelaunch_lg=%d,claim_trick=%d" % (rng.choice([16, 30]), rng.choice([0, 1, 1]))}
        env["GKRHIP_SPEC_LG"] = str(max(int(env["GKRHIP_SPEC_LG"]), int(env["GKRHIP_HOST_TAIL"]) + 1))   # the export round itself qualifies
        sizes = sorted(rng.sample(range(int(env["GKRHIP_HOST_TAIL"]) + 4, 15), 2))
        _run_case(env, ",".join(map(str, sizes)))


def test_error_while_a_prelaunched_kernel_waits(gk):
    """An error return between the pre-launch of a round and the publication of its challenge (injected by a test hook):
    the call fails with that error, the waiting kernel is told to leave, and the NEXT proofs on the same lane -- whose
    pre-launched kernels poll the same slot and mailbox -- are bit-exact again (the abort tags were cleared)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, sys
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import coracle as c
        gk = importlib.import_module("gkr-mimc_amd")
        gk.init(0)
        bn = 12
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
        qp = c.random_fr_array(bn)
        gk.set_option("test_fail_after_prelaunch", 3)      # fault injection is armed through the option, never the environment
        try:
            s.prove(qp)
            raise SystemExit("the injected failure did not surface")
        except gk.GkrHipError as e:
            assert "injected failure" in str(e), e
        i0 = c.random_fr_array(1 << bn)
        want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
        for _ in range(3):
            assert np.array_equal(s.prove(qp), want)
        assert gk.profile_get()["prelaunched_rounds"] > 0
        print("ABORT-PATH-OK")
    """ % (root, os.path.join(root, "oracle")))
    # the kernel left waiting is a pre-launched round kernel (GKRHIP_SPEC=0) or a speculative launch two rounds ahead
    for spec in ("0", "2"):
        env = dict(os.environ, GKRHIP_PRELAUNCH="2", GKRHIP_SPEC=spec)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ABORT-PATH-OK" in out.stdout, (spec, out.stdout + out.stderr)


def test_layer_retried_after_a_missed_challenge(gk):
    """A kernel that waits for its challenge gives up after a second (seen for real only with every latency path forced on for
    a dozen lanes at once: a polling kernel did not see a challenge the host had written).  The round loop then fails
    recoverably and the layer's rounds run once more with nothing queued ahead -- same transcript, the proof is merely late.
    Provoked with a test hook that withholds ONE challenge: from a pre-launched round kernel and from a speculative launch."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, sys
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import coracle as c
        gk = importlib.import_module("gkr-mimc_amd")
        gk.init(0)
        bn = 12
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
        qp = c.random_fr_array(bn)
        i0 = c.random_fr_array(1 << bn)
        want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
        gk.profile_reset(0)
        gk.set_option("test_drop_challenge", 3)
        for _ in range(3):
            assert np.array_equal(s.prove(qp), want)
        assert gk.profile_get()["chal_retries"] == 1, gk.profile_get()
        print("RETRY-OK")
    """ % (root, os.path.join(root, "oracle")))
    for spec in ("0", "2"):
        env = dict(os.environ, GKRHIP_PRELAUNCH="2", GKRHIP_SPEC=spec)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "RETRY-OK" in out.stdout, (spec, out.stdout + out.stderr)


def test_lookahead_round0(gk):
    """The q-independent products of a cipher layer's round 0 computed during the previous layer (k_cipher_pre on the
    look-ahead stream) and consumed by k_cipher_round_wide<false, ., true>: same transcript, early and late lane
    weights, MiMC and GMiMC circuits, and with the look-ahead switched off."""
    on = {"GKRHIP_PRE": "2", "GKRHIP_GMAX": "8", "GKRHIP_CASE_EXPECT": "lookahead_round0"}
    _run_case(on, "11,12,14,15")
    _run_case(dict(on, GKR_CASE_OPTIONS="wt_late_lj=99"), "11,13")
    _run_case(dict(on, GKRHIP_PRELAUNCH="0", GKR_CASE_OPTIONS="wt_late_lj=1,solo_boost=0"), "10,12")
    _run_case(dict(on, GKRHIP_HOST_TAIL="0"), "11,12", circuit="gmimc")
    _run_case({"GKRHIP_PRE": "0", "GKRHIP_GMAX": "8", "GKRHIP_CASE_EXPECT_NOT": "lookahead_round0"}, "11,13")
    # the defaults, alone on the GPU: below 2^22 entries round 0 runs AHEAD of its point during the previous layer's host tail
    # and the products are not computed (plan_rounds); from 2^22 on both (the bN = 22 / 24 digest tests take that path)
    _run_case({"GKRHIP_GMAX": "8", "GKRHIP_CASE_EXPECT": "ahead_round0,prelaunched_rounds", "GKRHIP_CASE_EXPECT_NOT": "lookahead_round0"}, "12")
    _run_case({"GKRHIP_GMAX": "8", "GKRHIP_AHEAD": "0", "GKRHIP_CASE_EXPECT": "lookahead_round0,prelaunched_rounds"}, "12")


def test_round0_ahead_of_its_point(gk):
    """Round 0 of a cipher layer queued by the layer before it, at the start of that layer's host tail, when the last t
    coordinates of its point do not exist yet (k_cipher_round_wide<false, true, ., true>: class sums over the low t index bits,
    contracted on the host): same transcript with early and late host tails (t = 2 .. 6), small and large thread budgets, with
    and without the products of k_cipher_pre, MiMC and GMiMC circuits, forced for every lane, and switched off."""
    on = {"GKRHIP_AHEAD": "2", "GKRHIP_CASE_EXPECT": "ahead_round0"}
    _run_case(dict(on, GKRHIP_GMAX="8"), "11,12,14,15")
    _run_case(dict(on, GKRHIP_GMAX="8", GKRHIP_PRE="2", GKRHIP_CASE_EXPECT="ahead_round0,lookahead_round0"), "11,13")
    _run_case(dict(on, GKRHIP_GMAX="10", GKRHIP_PRE="0"), "13,16")
    for h in ("1", "2", "3", "5"):
        _run_case(dict(on, GKRHIP_GMAX="8", GKRHIP_HOST_TAIL=h), "11,12")
    _run_case(dict(on, GKRHIP_HOST_TAIL="6", GKRHIP_CASE_EXPECT=""), "12,17")          # t = 7: not taken
    _run_case(dict(on, GKRHIP_GMAX="9", GKRHIP_PRELAUNCH="0", GKRHIP_SPEC="0", GKRHIP_COOP="0"), "12,14")
    _run_case(dict(on, GKRHIP_GMAX="8"), "11,12", circuit="gmimc")
    _run_case({"GKRHIP_AHEAD": "0", "GKRHIP_GMAX": "8", "GKRHIP_CASE_EXPECT_NOT": "ahead_round0"}, "11,13")
    _run_case({"GKRHIP_CASE_EXPECT": "ahead_round0,prelaunched_rounds"}, "19")   # the defaults, alone on the GPU


def test_round_kernel_deferred_reduction_carry_corners(gk):
    """Tables made of the carry-corner values (limbs of 0xFFFFFFFF, q-1, 0, 1) through gkr.Prove with a small
    thread budget, so that every lane accumulates many wide products of extreme operands."""
    gk.set_option("g_max", 8)
    try:
        for bn in (10, 13):
            i0, i1, qp = nasty(1 << bn, 3 * bn), nasty(1 << bn, 5 * bn + 1), c.random_fr_array(bn)
            flat, outs = gk.gkr_prove_mimc(i0, i1, qp)
            oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i1, qp)
            assert np.array_equal(flat, oflat) and np.array_equal(outs, oouts)
    finally:
        gk.set_option("g_max", 16)


# ---------------------------------------------------------------- gkr.Prove (MimcCircuit)
def test_gkr_golden(gk):
    for e in load("gkr_mimc.json"):
        bn = e["bn"]
        i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
        flat, outs = gk.gkr_prove_mimc(i0, i0.copy(), qp)
        assert fr_to_hex(flat) == e["flat"]
        assert fr_to_hex(outs) == e["outputs"]


def test_gkr_bn10_digest(gk):
    """BASELINE config 1 (bN = 10)."""
    d = load("gkr_mimc_bn10_digest.json")
    bn = d["bn"]
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    flat, outs = gk.gkr_prove_mimc(i0, i0.copy(), qp)
    assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == d["sha256_flat"]
    assert hashlib.sha256(outs.astype("<u8").tobytes()).hexdigest() == d["sha256_outputs"]
    assert c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp) == 0


@pytest.mark.parametrize("bn", [0, 1, 2, 4, 6, 9, 12, 14])
def test_gkr_vs_oracle(gk, bn):
    # gkr/gkr_test.go:14-78 with distinct inputs
    i0 = c.random_fr_array(1 << bn)
    i1 = nasty(1 << bn, bn) if bn else c.from_u64(9)
    qp = c.random_fr_array(bn)
    flat, outs = gk.gkr_prove_mimc(i0, i1, qp)
    oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i1, qp)
    assert np.array_equal(outs, oouts)
    assert np.array_equal(flat, oflat)
    assert c.gkr_verify_mimc(bn, flat, i0, i1, outs, qp) == 0


@pytest.mark.parametrize("bn", [20, 22, 24])
def test_gkr_baseline_sizes_match_oracle_digest(gk, bn):
    """BASELINE configs 2 and 3 (bN = 20, 24) and bN = 22: the full transcript and the output table equal the
    C oracle's, through SHA-256 digests the oracle produced on the host CPU (158 s for bN = 24 on 16 cores;
    tests/golden/gen_big_digests.py).  Inputs are generated on the device (RandomFrArray)."""
    want = [e for e in load("gkr_mimc_big_digests.json") if e["bn"] == bn][0]
    s = gk.MimcSession(bn)
    s.synth_inputs()
    s.assign()
    flat = s.prove(c.random_fr_array(bn))
    assert flat.shape[0] == want["n_elements"]
    assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == want["sha256_flat"]
    assert hashlib.sha256(s.outputs().astype("<u8").tobytes()).hexdigest() == want["sha256_outputs"]
    s.close()


def test_gkr_session_repeatable_and_claims_consistent(gk):
    """Prove never mutates the resident assignment; every input-layer claim equals
    Evaluate(layer table, point) (gkr/gkr_test.go:35-44) computed on the device."""
    bn = 11
    s = gk.MimcSession(bn)
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    s.load_inputs(i0, i0.copy())
    s.assign()
    f1 = s.prove(qp)
    f2 = s.prove(qp)
    assert np.array_equal(f1, f2)
    oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
    assert np.array_equal(f1, oflat)
    assert np.array_equal(s.outputs(), oouts)
    # flat layout: 822*bn coefficients, then claims (layer0, layer1, 91 of layer 2, ...), then points
    claims0 = f1[822 * bn]
    qp0 = f1[822 * bn + 183: 822 * bn + 183 + bn]
    assert np.array_equal(s.evaluate_layer(0, qp0)[0], claims0)
    s.close()


def test_concurrent_host_buffer_calls(gk):
    """Host-buffer entry points borrow a lane each: Fold, FoldedEqTable, sumcheck.Prove and the one-shot gkr.Prove from
    four host threads at once, every result bit-identical to the oracle's."""
    import threading
    bn = 10
    n = 1 << bn
    X = [c.random_fr_array(n), nasty(n, 3)]
    ark = c.from_u64(145646)
    qs = c.random_fr_array(bn).reshape(1, bn, 4)
    claims = c.evaluation(c.GATE_CIPHER, ark, qs, c.fr(0), X)
    r = c.mimc_hash(c.from_u64(9))
    want = {"fold": c.fold(X[1], r), "eq": c.folded_eq_table(qs[0]), "sc": c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims),
            "gkr": c.gkr_prove_mimc(bn, X[0], X[1], qs[0])[0]}
    errs = []

    def work(kind):
        try:
            for _ in range(6):
                if kind == "fold":
                    assert np.array_equal(gk.fold(X[1], r), want["fold"])
                elif kind == "eq":
                    assert np.array_equal(gk.folded_eq_table(qs[0]), want["eq"])
                elif kind == "sc":
                    for a, b in zip(gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark), want["sc"]):
                        assert np.array_equal(a, b)
                else:
                    assert np.array_equal(gk.gkr_prove_mimc(X[0], X[1], qs[0], want_outputs=False)[0], want["gkr"])
        except Exception as e:   # noqa: BLE001
            errs.append((kind, e))

    ths = [threading.Thread(target=work, args=(k,)) for k in ("fold", "eq", "sc", "gkr", "sc", "fold")]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs


def test_concurrent_sessions(gk):
    """Independent sessions own a lane (stream + buffers) each and may prove concurrently from different
    host threads; every proof is still bit-identical to the oracle."""
    import threading
    bns = [9, 11, 12]
    sessions, want = [], []
    for bn in bns:
        s = gk.MimcSession(bn)
        i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
        s.load_inputs(i0, i0.copy())
        s.assign()
        sessions.append((s, qp))
        want.append(c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0])
    got = [[None] * 3 for _ in bns]

    def work(k):
        s, qp = sessions[k]
        for rep in range(3):
            got[k][rep] = s.prove(qp)

    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(bns))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in range(len(bns)):
        for rep in range(3):
            assert np.array_equal(got[k][rep], want[k]), (bns[k], rep)
        sessions[k][0].close()


def test_gkr_synth_inputs_match_random_fr_array(gk):
    bn = 8
    s = gk.MimcSession(bn)
    s.synth_inputs()
    s.assign()
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    flat = s.prove(qp)
    oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
    assert np.array_equal(flat, oflat) and np.array_equal(s.outputs(), oouts)
    s.close()


def test_gkr_bn18_vs_oracle_and_bn20_verified(gk):
    """bN = 18: full transcript equality with the oracle.  bN = 20 (BASELINE config 2): the oracle's
    restated gkr.Verify accepts the GPU proof and a corrupted proof is rejected."""
    bn = 18
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    flat, outs = gk.gkr_prove_mimc(i0, i0.copy(), qp)
    oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
    assert np.array_equal(outs, oouts) and np.array_equal(flat, oflat)
    bn = 20
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    flat, outs = gk.gkr_prove_mimc(i0, i0.copy(), qp)
    assert c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp) == 0
    bad = flat.copy()
    bad[12345, 1] ^= np.uint64(4)
    assert c.gkr_verify_mimc(bn, bad, i0, i0, outs, qp) != 0


# ---------------------------------------------------------------- generic circuits: GMiMC (BASELINE config 5)
def _gmimc_inputs(n):
    return [o.random_fr_array(n), [(3 * i * i + 7) % o.Q for i in range(n)], [(i * i * i * 5 + 1) % o.Q for i in range(n)],
            [o.mimc_hash([i]) for i in range(n)]]


@pytest.mark.parametrize("bn", [0, 1, 3, 5])
def test_gmimc_circuit_vs_oracle(gk, bn):
    """gkr.Prove on the build-defined GMiMC (t = 2) circuit (add, cipher and copy layers): the transcript equals
    the Python oracle's, the outputs equal hash.GMimcHasher's compression of every instance."""
    n = 1 << bn
    ins = _gmimc_inputs(n)
    circ = o.gmimc_t2_circuit()
    a = o.assign(circ, *ins)
    qp = o.random_fr_array(bn)
    want = o.gkr_proof_to_vec(o.gkr_prove(circ, a, qp))
    s = gk.MimcSession(bn, layers=gk.gmimc_t2_circuit())
    assert s.num_inputs == 4 and s.proof_len == len(want)
    for k in range(4):
        s.load_input(k, c.from_ints(ins[k]))
    s.assign()
    flat = s.prove(c.from_ints(qp))
    assert c.to_ints(flat) == want
    assert c.to_ints(s.outputs()) == [o.gmimc_update([ins[0][k], ins[1][k]], [ins[2][k], ins[3][k]])[0] for k in range(n)]
    assert s.verify(c.from_ints(qp), flat)
    s.close()


@pytest.mark.parametrize("bn", [8, 12, 14])
def test_gmimc_circuit_vs_c_oracle(gk, bn):
    """Full-transcript equality with the C oracle's generic-circuit prover at sizes Python cannot reach."""
    n = 1 << bn
    circ = o.gmimc_t2_circuit()
    descs = c.circuit_descs(circ)
    ins = [c.random_fr_array(n), nasty(n, bn + 1), c.from_ints([int(v) for v in np.random.default_rng(bn).integers(0, 1 << 62, n)]),
           nasty(n, bn + 2)]
    qp = c.random_fr_array(bn)
    want, wouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
    s = gk.MimcSession(bn, layers=gk.gmimc_t2_circuit())
    for k in range(4):
        s.load_input(k, ins[k])
    s.assign()
    flat = s.prove(qp)
    assert np.array_equal(flat, want) and np.array_equal(s.outputs(), wouts)
    assert c.gkr_verify_circuit(descs, bn, flat, ins, wouts, qp) == 0
    s.close()


def test_gmimc_circuit_larger_sizes(gk):
    """bN = 10: outputs equal the reference hasher on every instance; bN = 14 and 22 (BASELINE config 5): the
    native gkr.Verify accepts the proof against the resident tables and rejects a corrupted one."""
    bn = 10
    n = 1 << bn
    ins = _gmimc_inputs(n)
    s = gk.MimcSession(bn, layers=gk.gmimc_t2_circuit())
    for k in range(4):
        s.load_input(k, c.from_ints(ins[k]))
    s.assign()
    assert c.to_ints(s.outputs()) == [o.gmimc_update([ins[0][k], ins[1][k]], [ins[2][k], ins[3][k]])[0] for k in range(n)]
    qp = c.random_fr_array(bn)
    flat = s.prove(qp)
    assert s.verify(qp, flat)
    s.close()
    for bn in (14, 22):
        s = gk.MimcSession(bn, layers=gk.gmimc_t2_circuit())
        s.synth_inputs()
        s.assign()
        qp = c.random_fr_array(bn)
        flat = s.prove(qp)
        assert np.array_equal(flat, s.prove(qp))
        assert s.verify(qp, flat)
        bad = flat.copy()
        bad[len(bad) // 2, 0] ^= np.uint64(1)
        assert not s.verify(qp, bad)
        s.close()


# ---------------------------------------------------------------- the gate table (circuit.Gate plug point)
def _variadic_circuit():
    """A layered circuit over registered gates of 1, 3 and 4 inputs (the same one tests/test_oracle.py proves with
    both restatements): returns (pyoracle circuit, library layer list)."""
    L = [o.Layer([]) for _ in range(4)]
    L.append(o.Layer([0], o.IdentityGate()))
    L.append(o.Layer([4, 1, 2], o.SumGate(o.ARKS[0], 7)))
    L.append(o.Layer([5, 4, 3], o.SumGate(o.ARKS[1], 1)))
    L.append(o.Layer([6], o.IdentityGate()))
    L.append(o.Layer([7], o.SumGate(5, 7)))
    L.append(o.Layer([7, 8, 5, 6], o.SumGate(o.ARKS[2], 7)))
    circ = o.build_circuit(L)
    return circ, _library_layers(circ)


def _library_layers(circ):
    def gate_id(lay):
        g = lay.gate
        if g is None:
            return -1
        if g.kind in ("sum", "sum_pow7"):
            return importlib.import_module("gkr-mimc_amd").gate_register(
                "test-%s-%d" % (g.kind, len(lay.In)), len(lay.In), (1 << len(lay.In)) - 1, g.power)
        return {"identity": 0, "cipher": 1, "add": 2}[g.kind]
    return [(gate_id(l), list(l.In), None if l.gate is None else o.to_mont_limbs(getattr(l.gate, "ark", 0))) for l in circ]


@pytest.mark.parametrize("arity,power", [(1, 7), (3, 1), (3, 7), (4, 1), (4, 7)])
def test_registered_gates_eval_and_sumcheck_vs_oracle(gk, arity, power):
    """Gate.EvalBatch and sumcheck.Prove for registered descriptors (variadic gates, circuit/gates.go:16-18) against
    the C oracle's restated gates: one claim (fused or reference-shaped rounds) and several claims."""
    gate = gk.gate_register("test-%s-%d" % ("sum" if power == 1 else "sum_pow7", arity), arity, (1 << arity) - 1, power)
    ogate = c.GATE_SUM if power == 1 else c.GATE_SUM_POW7
    ark = c.from_ints([o.ARKS[7]])
    for bn in (1, 4, 9):
        n = 1 << bn
        X = [nasty(n, 10 * bn + k) if k % 2 else c.random_fr_array(n) for k in range(arity)]
        assert np.array_equal(gk.gate_eval_batch(gate, ark, X), c.gate_eval_batch(ogate, ark, X))
        for nq in (1, 3):
            qs = np.stack([c.from_ints([(7 * i * j + i + 3) % o.Q for j in range(bn)]) for i in range(nq)])
            claims = np.concatenate([c.evaluation(ogate, ark, qs[i:i + 1], c.fr(0), X) for i in range(nq)])
            got = gk.sumcheck_prove(X, qs, claims, gate, ark)
            want = c.sumcheck_prove(ogate, ark, X, qs, claims)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (arity, power, bn, nq)


@pytest.mark.parametrize("bn", [0, 1, 4, 10])
def test_circuit_of_registered_gates_vs_oracle(gk, bn):
    """gkr.Prove over a circuit with 1-, 3- and 4-input gates: transcript and outputs equal the C oracle's, the native
    verifier (session and host-table forms) and the oracle's verifier accept, a corrupted proof is rejected."""
    circ, layers = _variadic_circuit()
    descs = c.circuit_descs(circ)
    n = 1 << bn
    ins = [c.random_fr_array(n), nasty(n, bn + 5), c.from_ints([(3 * j * j + 1) % 1000003 for j in range(n)]), nasty(n, bn + 6)]
    qp = c.random_fr_array(bn)
    want, wouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
    s = gk.MimcSession(bn, layers=layers)
    for k in range(4):
        s.load_input(k, ins[k])
    s.assign()
    flat = s.prove(qp)
    assert np.array_equal(flat, want) and np.array_equal(s.outputs(), wouts)
    assert s.verify(qp, flat)
    assert gk.gkr_verify(layers, flat, ins, wouts, qp)
    assert c.gkr_verify_circuit(descs, bn, flat, ins, wouts, qp) == 0
    f2, o2 = gk.gkr_prove(layers, ins, qp)                 # the one-call form on host tables
    assert np.array_equal(f2, want) and np.array_equal(o2, wouts)
    if bn:
        bad = flat.copy()
        bad[len(bad) // 3, 2] ^= np.uint64(16)
        assert not gk.gkr_verify(layers, bad, ins, wouts, qp)
    s.close()


def _random_circuit(rng):
    """A random layered circuit over the gate family: 2-4 input layers, 4-10 gate layers of 1-4 inputs drawn from
    earlier layers, explicit copy layers for everything that is used more than once (circuit/circuit.go:36-41), the
    unused layers pruned so that only the last layer has no consumer."""
    n_in = int(rng.integers(2, 5))
    raw = [([], None)] * n_in
    for _ in range(int(rng.integers(4, 11))):
        k = min(int(rng.integers(1, 5)), len(raw))
        ins = [int(v) for v in rng.choice(len(raw), size=k, replace=False)]   # distinct: Out is searched by value (gkr/prover.go:78-86)
        power = 7 if rng.random() < 0.5 else 1
        raw.append((ins, (int(rng.integers(0, 1 << 62)), power)))
    # keep what reaches the last layer
    need = {len(raw) - 1}
    for l in range(len(raw) - 1, -1, -1):
        if l in need:
            need.update(raw[l][0])
    need.update(range(n_in))
    # every use of a layer that has several uses goes through a copy layer of its own... the reference's rule is weaker
    # (only INPUT layers need it, circuit/circuit.go:36-41), so do exactly that: copies for multiply-used inputs
    uses = {l: 0 for l in range(len(raw))}
    for l in sorted(need):
        for p in raw[l][0]:
            uses[p] += 1
    L, ren = [], {}
    for l in range(n_in):
        L.append(o.Layer([]))
        ren[l] = l
    for l in range(n_in):
        if uses[l] == 0:                      # an input nothing reads: give it a consumer (input layers need a claim)
            raw.append(([l, len(raw) - 1], (0, 1)))
            need.add(len(raw) - 1)
            uses[l] += 1
        if uses[l] > 1:
            L.append(o.Layer([l], o.IdentityGate()))
            ren[l] = len(L) - 1
    for l in range(n_in, len(raw)):
        if l not in need:
            continue
        ins, (ark, power) = raw[l]
        L.append(o.Layer([ren[p] for p in ins], o.SumGate(ark, power)))
        ren[l] = len(L) - 1
    return o.build_circuit(L), n_in


@pytest.mark.parametrize("seed", range(8))
def test_random_circuits_vs_oracle(gk, seed):
    """gkr.Prove on random circuits of registered gates (1-4 inputs, power 1 or 7, layers with several consumers, i.e.
    multi-claim sumchecks of every gate shape): transcript, outputs and verifier verdicts equal the C oracle's."""
    rng = np.random.default_rng(1000 + seed)
    circ, n_in = _random_circuit(rng)
    layers = _library_layers(circ)
    descs = c.circuit_descs(circ)
    for bn in (1, 3, 7):
        n = 1 << bn
        ins = [nasty(n, seed * 10 + k) if k % 2 else c.from_ints([int(v) for v in rng.integers(0, 1 << 62, n)]) for k in range(n_in)]
        qp = c.random_fr_array(bn)
        want, wouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
        assert c.gkr_verify_circuit(descs, bn, want, ins, wouts, qp) == 0
        s = gk.MimcSession(bn, layers=layers)
        for k in range(n_in):
            s.load_input(k, ins[k])
        s.assign()
        flat = s.prove(qp)
        assert np.array_equal(flat, want), (seed, bn)
        assert np.array_equal(s.outputs(), wouts)
        assert gk.gkr_verify(layers, flat, ins, wouts, qp)
        s.close()


@pytest.mark.parametrize("t", [4, 8])
def test_gmimc_t4_t8_circuits_vs_oracle(gk, t):
    """GMiMC for t = 4 and 8 (hash/gmimc.go:16-20): the library's circuit (three-input feed-forward gate) against the C
    oracle's transcript; outputs against the reference hasher's compression on every instance."""
    layers, imap = gk.gmimc_circuit(t)
    circ, ref_map = o.gmimc_circuit(t)
    assert imap == ref_map
    descs = c.circuit_descs(circ)
    for bn in (2, 8):
        n = 1 << bn
        rng = np.random.default_rng(100 * t + bn)
        vals = [[int(v) for v in rng.integers(0, 1 << 62, n)] for _ in range(2 * t)]
        ins = [c.from_ints(vals[j]) for j in imap]
        qp = c.random_fr_array(bn)
        want, wouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
        s = gk.MimcSession(bn, layers=layers)
        for k, tab in enumerate(ins):
            s.load_input(k, tab)
        s.assign()
        flat = s.prove(qp)
        assert np.array_equal(flat, want) and np.array_equal(s.outputs(), wouts)
        assert gk.gkr_verify(layers, flat, ins, wouts, qp)
        if bn == 2:
            assert c.to_ints(wouts) == [o.gmimc_update([vals[j][k] for j in range(t)], [vals[t + j][k] for j in range(t)])[0]
                                        for k in range(n)]
        s.close()


@pytest.mark.parametrize("t,nb", [(2, 2), (4, 2), (2, 3)])
def test_gmimc_sponge_circuit_vs_oracle(gk, t, nb):
    """The whole sponge hash.GMimcHasher.Hash (hash/gmimc.go:29-49) over nb blocks as ONE circuit (state carried from
    block to block, layers with two consumers, one-input gates in the first block): transcript against the C oracle,
    outputs against the reference hasher's digest of every instance's message."""
    layers, imap = gk.gmimc_hash_circuit(t, nb)
    circ, ref_map = o.gmimc_hash_circuit(t, nb)
    assert imap == ref_map
    descs = c.circuit_descs(circ)
    for bn in (2, 7):
        n = 1 << bn
        rng = np.random.default_rng(1000 * t + 10 * nb + bn)
        vals = [[int(v) for v in rng.integers(0, 1 << 62, n)] for _ in range(t * nb)]
        ins = [c.from_ints(vals[j]) for j in imap]
        qp = c.random_fr_array(bn)
        want, wouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
        s = gk.MimcSession(bn, layers=layers)
        for k, tab in enumerate(ins):
            s.load_input(k, tab)
        s.assign()
        flat = s.prove(qp)
        assert np.array_equal(flat, want) and np.array_equal(s.outputs(), wouts)
        assert gk.gkr_verify(layers, flat, ins, wouts, qp)
        if bn == 2:
            assert c.to_ints(wouts) == [o.gmimc_hash([vals[j][k] for j in range(t * nb)], t) for k in range(n)]
        s.close()


@pytest.mark.parametrize("t,bn", [(4, 16), (4, 20), (8, 16)])
def test_gmimc_t4_t8_match_oracle_digest(gk, t, bn):
    """The GMiMC circuits with the registered three-input feed-forward gate at larger sizes: SHA-256 of the transcript and
    of the output table against the C oracle's (tests/golden/gkr_gmimc_t48_digests.json; every input layer =
    RandomFrArray(2^bN), generated on the device)."""
    want = [e for e in load("gkr_gmimc_t48_digests.json") if e["t"] == t and e["bn"] == bn][0]
    layers, _imap = gk.gmimc_circuit(t)
    assert len(layers) == want["n_layers"]
    s = gk.MimcSession(bn, layers=layers)
    s.synth_inputs()
    s.assign()
    flat = s.prove(c.random_fr_array(bn))
    assert flat.shape[0] == want["n_elements"]
    assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == want["sha256_flat"]
    assert hashlib.sha256(s.outputs().astype("<u8").tobytes()).hexdigest() == want["sha256_outputs"]
    s.close()


@pytest.mark.parametrize("bn", [14, 20, 22])
def test_gmimc_baseline_sizes_match_oracle_digest(gk, bn):
    """BASELINE config 5 (bN = 22) and two smaller sizes: SHA-256 of the GPU transcript and of the output table
    against the C oracle's (tests/golden/gkr_gmimc_big_digests.json, tests/golden/gen_big_digests.py gmimc)."""
    want = [e for e in load("gkr_gmimc_big_digests.json") if e["bn"] == bn][0]
    s = gk.MimcSession(bn, layers=gk.gmimc_t2_circuit())
    s.synth_inputs()
    s.assign()
    flat = s.prove(c.random_fr_array(bn))
    assert flat.shape[0] == want["n_elements"]
    assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == want["sha256_flat"]
    assert hashlib.sha256(s.outputs().astype("<u8").tobytes()).hexdigest() == want["sha256_outputs"]
    s.close()


def _split_flat(circ_out_sizes, degs, bn, flat):
    """flat proof (GkrProofToVec order, hints.go:236-271) -> (claims[layer][slot], qprimes[layer][slot])."""
    L = len(circ_out_sizes)
    cur = sum(bn * (d + 2) for d in degs if d is not None)
    claims, qps = [], []
    for l in range(L):
        claims.append(flat[cur:cur + circ_out_sizes[l]])
        cur += circ_out_sizes[l]
    for l in range(L):
        slots = 1 if l == L - 1 else circ_out_sizes[l]
        qps.append(flat[cur:cur + slots * bn].reshape(slots, bn, 4))
        cur += slots * bn
    assert cur == flat.shape[0]
    return claims, qps


def test_gkr_all_claims_equal_evaluate_bn20(gk):
    """gkr/gkr_test.go:35-44 at BASELINE config 2's size: EVERY claim of the proof (183 of them: 91 on the key
    copy, one per cipher layer, the two inputs) equals MultiLin.Evaluate of that layer's table at the claim's point,
    evaluated on the resident (never mutated) assignment."""
    bn = 20
    circ = o.mimc_circuit()
    outs = [len(l.Out) for l in circ]
    degs = [None if l.gate is None else l.gate.degree() for l in circ]
    s = gk.MimcSession(bn)
    s.synth_inputs()
    s.assign()
    flat = s.prove(c.random_fr_array(bn))
    claims, qps = _split_flat(outs, degs, bn, flat)
    checked = 0
    for l in range(len(circ) - 1):
        for w in range(outs[l]):
            assert np.array_equal(s.evaluate_layer(l, qps[l][w]), claims[l][w:w + 1]), (l, w)
            checked += 1
    assert checked == 183
    s.close()


def test_gkr_bn22_accepted_by_oracle_verifier(gk):
    """The ORACLE's restated gkr.Verify (not the library's own) on the GPU's bN = 22 proof: inputs are regenerated
    on the host, the output table comes from the device and is itself pinned by the committed digest."""
    bn = 22
    want = [e for e in load("gkr_mimc_big_digests.json") if e["bn"] == bn][0]
    s = gk.MimcSession(bn)
    s.synth_inputs()
    s.assign()
    qp = c.random_fr_array(bn)
    flat = s.prove(qp)
    outs = s.outputs()
    s.close()
    assert hashlib.sha256(outs.astype("<u8").tobytes()).hexdigest() == want["sha256_outputs"]
    i0 = c.random_fr_array(1 << bn)
    assert c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp) == 0
    bad = flat.copy()
    bad[12345, 1] ^= np.uint64(4)
    assert c.gkr_verify_mimc(bn, bad, i0, i0, outs, qp) != 0


@pytest.mark.parametrize("bn", [20, 22])
def test_oneshot_upload_and_assignment_in_slices(gk, bn):
    """gkrhip_gkr_prove_mimc on host buffers from 2^20 entries on: the inputs cross PCIe in slices on the lane's second stream
    while the 91 element-wise layers of Circuit.Assign run on the slices that have landed (4 slices at bN = 20, 8 at 22).
    Transcript and output table against the committed digests of the C oracle; in1 != in0 and the regular-form variant
    against the session path (whole-table upload, whole-table layers)."""
    want = [e for e in load("gkr_mimc_big_digests.json") if e["bn"] == bn][0]
    i0 = c.random_fr_array(1 << bn)
    qp = c.random_fr_array(bn)
    flat, outs = gk.gkr_prove_mimc(i0, i0.copy(), qp)
    assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == want["sha256_flat"]
    assert hashlib.sha256(outs.astype("<u8").tobytes()).hexdigest() == want["sha256_outputs"]
    if bn > 20:
        return
    i1 = i0[::-1].copy()
    i1[12345] = nasty(1, 7)[0]
    s = gk.MimcSession(bn)
    s.load_inputs(i0, i1)
    s.assign()
    sflat, souts = s.prove(qp), s.outputs()
    s.close()
    flat, outs = gk.gkr_prove_mimc(i0, i1, qp)
    assert np.array_equal(flat, sflat) and np.array_equal(outs, souts)
    bad = i1.copy()
    bad[(1 << bn) - 3] = np.array([0xFFFFFFFFFFFFFFFF] * 4, dtype=np.uint64)      # not a canonical element, in the last slice
    with pytest.raises(gk.GkrHipError):
        gk.gkr_prove_mimc(i0, bad, qp)
    flat2, _ = gk.gkr_prove_mimc(i0, i1, qp)                                        # the lane is usable afterwards
    assert np.array_equal(flat2, sflat)


# ---------------------------------------------------------------- verifier and wire-format helpers
@pytest.mark.parametrize("bn", [0, 1, 3, 8, 13])
def test_native_verifier_agrees_with_oracle(gk, bn):
    """gkr.Verify (gkr/verifier.go): accepts the prover's output, rejects every single-element corruption we
    try, and agrees with the oracle's restated verifier."""
    i0 = c.random_fr_array(1 << bn)
    i1 = nasty(1 << bn, bn + 40) if bn else c.from_u64(5)
    qp = c.random_fr_array(bn)
    flat, outs = gk.gkr_prove_mimc(i0, i1, qp)
    assert gk.gkr_verify_mimc(flat, i0, i1, outs, qp)
    assert c.gkr_verify_mimc(bn, flat, i0, i1, outs, qp) == 0
    rng = np.random.default_rng(bn)
    if bn:
        for pos in rng.integers(0, flat.shape[0], 6):
            bad = flat.copy()
            bad[pos, int(rng.integers(0, 4))] ^= np.uint64(1 << int(rng.integers(0, 60)))
            ours = gk.gkr_verify_mimc(bad, i0, i1, outs, qp)
            theirs = c.gkr_verify_mimc(bn, bad, i0, i1, outs, qp) == 0
            assert ours == theirs and not ours
    bad_out = outs.copy()
    bad_out[0, 0] ^= np.uint64(2)
    assert not gk.gkr_verify_mimc(flat, i0, i1, bad_out, qp)
    bad_in = i1.copy()
    bad_in[0, 1] ^= np.uint64(8)          # still a canonical element
    assert not gk.gkr_verify_mimc(flat, i0, bad_in, outs, qp)


@pytest.mark.parametrize("bn", [0, 1, 5, 11])
def test_gkr_prove_mimc_on_regular_form_buffers(gk, bn):
    """gkrhip_gkr_prove_mimc_regular: the hint's body on big.Int words (prover/gadget/hints.go:197-233).  Inputs, qPrime,
    the flat proof and the output table in REGULAR form equal the Montgomery-form call's, element for element, after
    conversion; the oracle's transcript pins both."""
    n = 1 << bn
    rng = np.random.default_rng(77 + bn)

    def rnd(k):
        return [int.from_bytes(rng.bytes(32), "little") % o.Q for _ in range(k)]

    def words(vals):
        return np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in vals], dtype=np.uint64).reshape(-1, 4)

    v0, v1, vq = rnd(n), rnd(n), rnd(bn)
    flat_r, outs_r = gk.gkr_prove_mimc(words(v0), words(v1), words(vq), regular=True)
    want, wouts, _ = c.gkr_prove_mimc(bn, c.from_ints(v0), c.from_ints(v1), c.from_ints(vq) if bn else c.from_ints([]))
    assert np.array_equal(flat_r, words(c.to_ints(want)))
    assert np.array_equal(outs_r, words(c.to_ints(wouts)))
    flat_m, outs_m = gk.gkr_prove_mimc(c.from_ints(v0), c.from_ints(v1), c.from_ints(vq) if bn else c.from_ints([]))
    assert np.array_equal(flat_m, want) and np.array_equal(outs_m, wouts)
    # a value that is not below q is refused
    bad = words(v0)
    bad[0] = words([o.Q])[0]
    with pytest.raises(gk.prover.GkrHipError):
        gk.gkr_prove_mimc(bad, words(v1), words(vq), regular=True)


def test_non_canonical_input_is_refused(gk):
    """gnark-crypto keeps fr.Element below q and the round kernels' lazy-reduction bounds rely on it: a table with
    an element >= q is refused at the boundary instead of yielding silently different sums."""
    t = c.random_fr_array(8)
    t[5] = [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]   # q itself
    with pytest.raises(gk.prover.GkrHipError, match="canonical"):
        gk.fold(t, c.from_u64(5))
    t[5] = [0xFFFFFFFFFFFFFFFF] * 4
    with pytest.raises(gk.prover.GkrHipError, match="canonical"):
        gk.evaluate(t, c.random_fr_array(3))
    assert np.array_equal(gk.fold(c.random_fr_array(8), c.from_u64(5)), c.fold(c.random_fr_array(8), c.from_u64(5)))


def test_session_verify_full_size(gk):
    """BASELINE config 3 size: the bN = 24 proof is accepted by gkr.Verify run against the resident tables,
    and a corrupted transcript is rejected."""
    bn = 24
    s = gk.MimcSession(bn)
    s.synth_inputs()
    s.assign()
    qp = c.random_fr_array(bn)
    flat = s.prove(qp)
    assert s.verify(qp, flat)
    bad = flat.copy()
    bad[777, 2] ^= np.uint64(1)
    assert not s.verify(qp, bad)
    s.close()


def test_wire_format_helpers(gk):
    # ToBigIntRegular / SetBigInt for a slice (prover/gadget/hints.go:202-205,236-271)
    vals = [0, 1, 12, o.Q - 1, 1 << 200, 1808205620575546259657963589762746470347087906694759866517376279978241663265]
    mont = c.from_ints(vals)
    reg = gk.to_regular(mont)
    assert [sum(int(reg[i, k]) << (64 * k) for k in range(4)) for i in range(len(vals))] == vals
    assert np.array_equal(gk.from_regular(reg), mont)
    big = nasty(1 << 12, 77)
    assert c.to_ints(gk.from_regular(gk.to_regular(big))) == c.to_ints(big)
    # HashHint.Call for a batch: MimcKeyedPermutation(block, state)  (hints.go:134-145)
    x, key = nasty(300, 5), c.random_fr_array(300)
    got = gk.mimc_permutation_batch(x, key)
    for i in (0, 1, 2, 17, 299):
        assert np.array_equal(got[i], c.mimc_keyed_permutation(x[i:i + 1], key[i:i + 1])[0])
    for e in load("kat.json")["mimc_perm"]:
        assert fr_to_hex(gk.mimc_permutation_batch(hex_to_fr(e["x"]), hex_to_fr(e["key"]))) == [e["out"]]


# ---------------------------------------------------------------- compiled caller (what a cgo shim does)
def test_cpp_abi_harness(gk, tmp_path):
    """tests/cpp/test_abi_gkr.cpp: the reference's TestGKR (gkr/gkr_test.go:14-78) and the poly / sumcheck entry
    points driven from compiled code through include/gkrhip.h, checked against the oracle library."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s"])
    exe = str(tmp_path / "test_abi_gkr")
    lib, orc = os.path.join(root, "gkr-mimc_amd"), os.path.join(root, "oracle")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "test_abi_gkr.cpp"),
                           "-L" + lib, "-lgkrhip", "-L" + orc, "-lgkr_oracle", "-Wl,-rpath," + lib, "-Wl,-rpath," + orc,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-fopenmp"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "abi-harness fails=0" in out.stdout, out.stdout + out.stderr


def test_soak_lanes_deterministic(gk):
    """tools/stress.py for 20 s: 14 lanes proving sizes 2^9..2^20 concurrently; every proof is byte-identical to the
    first of its size and verifies (races in the hand-off flags or the atomic accumulation would show up here)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress.py"), "20"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "mismatches: []" in out.stdout, out.stdout + out.stderr


def test_soak_one_proof_at_a_time(gk):
    """The paths a proof takes when it is alone on the GPU, as they are by default: proofs of fourteen sizes one at a time for a
    few seconds, every one byte-identical to the first of its size (which the native verifier accepted), and the speculative,
    cooperative, pre-launched and look-ahead paths all taken."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_solo.py"), "8"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: []" in out.stdout, out.stdout + out.stderr
    counts = dict(zip(("spec", "coop", "pre", "look"), [int(w.strip(",")) for w in out.stdout.split("speculative rounds")[1].replace(
        "cooperative", "").replace("pre-launched", "").replace("look-ahead", "").split()[:4]]))
    assert all(v > 0 for v in counts.values()), counts


def test_twelve_lanes_of_one_size_with_the_solo_paths_forced_on(gk):
    """Twelve lanes proving bN = 18 at once with the look-ahead kernel and the pre-launched rounds forced on, for the thread caps
    2^15 (the library's choice from ten proofs in flight) and 2^16: the load under which the look-ahead kernel's former
    lowest-priority stream left incomplete products (4 % wrong proofs, tools/stress_one_size.py); now every proof is byte-equal."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for gmax in ("15", "16", None):
        env = dict(os.environ, GKRHIP_PRELAUNCH="2", GKRHIP_PRE="2", GKRHIP_COOP="2", GKRHIP_SPEC="0")
        if gmax:
            env["GKRHIP_GMAX"] = gmax
        for rep in range(2):
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_one_size.py"), "18", "12", "20"], capture_output=True,
                                 text=True, timeout=600, env=env)
            assert out.returncode == 0 and "mismatches (lane, proof, last differing row): []" in out.stdout, (gmax, rep, out.stdout + out.stderr)
    # the same load with the library's own choices (no path forced), and twenty-four lanes of the size of BASELINE config 2
    for bn, lanes, per in (("18", "12", "20"), ("20", "24", "6")):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_one_size.py"), bn, lanes, per], capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0 and "mismatches (lane, proof, last differing row): []" in out.stdout, (bn, out.stdout + out.stderr)


def test_soak_lanes_with_the_solo_paths_forced_on(gk):
    """The same soak with the round-3 serial-latency paths forced on for EVERY lane (pre-launched rounds polling their
    challenge, look-ahead kernels on second streams, the cooperative kernel): many spinning kernels, look-ahead launches and
    mailboxes at once must neither deadlock (the in-kernel wait would time out) nor change a byte of any proof."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # GKRHIP_SPEC=0: the cooperative kernel takes the small rounds; 2: the speculative launches (two kernels queued per lane, three
    # challenge slots and mailboxes per lane) take them
    for spec, secs in (("0", "15"), ("2", "12")):
        env = dict(os.environ, GKRHIP_PRELAUNCH="2", GKRHIP_PRE="2", GKRHIP_COOP="2", GKRHIP_SPEC=spec)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress.py"), secs], capture_output=True, text=True,
                             timeout=900, env=env)
        assert out.returncode == 0 and "mismatches: []" in out.stdout, (spec, out.stdout + out.stderr)
