"""CPU tests: pin the oracle (C restatement + Python big-int restatement) on the reference's known
answers, on the committed golden vectors, and on the reference's self-consistency tests restated.

Mirrors: hash/hash_test.go, poly/multilin_test.go, poly/eq_test.go, poly/lagrange_test.go,
sumcheck/prover_test.go (genericTest), gkr/gkr_test.go (TestGKR), examples/mimc_test.go."""
import hashlib

import numpy as np
import pytest

import coracle as c
import pyoracle as o
from util import fr_to_hex, hex_to_fr, load


def test_mimc_case_kat():
    # hash/hash_test.go:21-27
    want = 1808205620575546259657963589762746470347087906694759866517376279978241663265
    assert o.mimc_hash([12]) == want
    assert c.to_ints(c.mimc_hash(c.from_ints([12])))[0] == want


def test_montgomery_layout():
    # Montgomery(12) limbs
    want = [0x4b649097efffffc1, 0x1b39d62a0b5d4c40, 0xa43ed816212b4113, 0x1750b1ba94c995bb]
    assert [int(v) for v in c.from_u64(12)[0]] == want
    assert o.to_mont_limbs(12) == want


def test_fold_kat():
    # poly/multilin_test.go:12-31
    out = c.fold(c.from_ints([0, 1, 2, 3]), c.from_u64(5))
    assert c.to_ints(out) == [10, 11]


def test_fold_chunk_equals_whole():
    # poly/multilin_test.go:33-53 (chunked == whole): the C fold is chunked over OpenMP tasks
    t = c.random_fr_array(1 << 13)
    r = c.mimc_hash(c.from_u64(3))
    assert c.to_ints(c.fold(t, r)) == o.fold(o.random_fr_array(1 << 13), c.to_ints(r)[0])


def test_lagrange_kat():
    # poly/lagrange_test.go:10-29
    L = c.lagrange_coefficient(7)[2]
    for i in range(7):
        y = c.to_ints(c.eval_univariate(L, c.from_u64(i)))[0]
        assert y == (1 if i == 2 else 0)


def test_univariate_kat():
    # snark/polynomial/univariate_test.go:39-45: X^3+2X^2+3X+4
    p = c.from_ints([4, 3, 2, 1])
    s = (c.to_ints(c.eval_univariate(p, c.from_u64(0)))[0] + c.to_ints(c.eval_univariate(p, c.from_u64(1)))[0]) % o.Q
    assert s == 14
    assert c.to_ints(c.eval_univariate(p, c.from_u64(5)))[0] == 194


def test_golden_kat_file():
    k = load("kat.json")
    for e in k["mimc_hash"]:
        assert fr_to_hex(c.mimc_hash(hex_to_fr(e["in"]))) == [e["out"]]
        assert o.to_hex(o.mimc_hash([o.from_hex(h) for h in e["in"]])) == e["out"]
    for e in k["mimc_perm"]:
        assert fr_to_hex(c.mimc_keyed_permutation(hex_to_fr(e["x"]), hex_to_fr(e["key"]))) == [e["out"]]
    assert fr_to_hex(c.random_fr_array(16)) == k["random_fr_array_16"]
    assert fr_to_hex(c.lagrange_coefficient(9).reshape(-1, 4)) == sum(k["lagrange_9"], [])
    for e in k["gmimc_hash"]:
        assert o.to_hex(o.gmimc_hash([o.from_hex(h) for h in e["in"]], e["t"])) == e["out"]


def test_golden_poly_file():
    p = load("poly.json")
    for e in p["fold"]:
        assert fr_to_hex(c.fold(hex_to_fr(e["tbl"]), hex_to_fr(e["r"]))) == e["out"]
    for e in p["eq"]:
        q = hex_to_fr(e["q"])
        m = hex_to_fr(e["mult"]) if e["mult"] else None
        assert fr_to_hex(c.folded_eq_table(q, m)) == e["out"]
    for e in p["eval_eq"]:
        assert fr_to_hex(c.eval_eq(hex_to_fr(e["q"]), hex_to_fr(e["h"]))) == [e["out"]]


@pytest.mark.parametrize("bn", range(0, 13))
def test_eq_table_vs_eval_eq(bn):
    # poly/eq_test.go:12-26: EvalEq(q,h) == FoldedEqTable(q).Evaluate(h)
    q, h = c.random_fr_array(bn), c.random_fr_array(bn)
    assert (c.evaluate(c.folded_eq_table(q), h) == c.eval_eq(q, h)).all()


@pytest.mark.parametrize("bn", range(2, 12))
def test_eq_table_chunks(bn):
    # poly/eq_test.go:28-58
    q = c.random_fr_array(bn)
    whole = c.folded_eq_table(q)
    for lg in range(1, bn):
        assert (c.chunked_eq_table(q, 1 << lg) == whole).all()


def _golden_sumcheck():
    return load("sumcheck.json")


def test_golden_sumcheck_file():
    for e in _golden_sumcheck():
        bn = e["bn"]
        n = 1 << bn
        X = [c.from_ints(range(n)), c.from_ints(range(n))]
        qs = np.stack([hex_to_fr(q).reshape(bn, 4) for q in e["qprimes"]]) if bn else np.zeros((len(e["qprimes"]), 0, 4), np.uint64)
        gate = c.GATE_CIPHER if e["kind"] == "cipher" else c.GATE_IDENTITY
        ark = hex_to_fr(e["ark"]) if e["ark"] else None
        proof, chal, final = c.sumcheck_prove(gate, ark, X, qs, hex_to_fr(e["claims"]))
        assert [fr_to_hex(r) for r in proof] == e["proof"]
        assert fr_to_hex(chal) == e["challenges"]
        assert fr_to_hex(final) == e["final"]


@pytest.mark.parametrize("bn", range(0, 13))
@pytest.mark.parametrize("kind", ["cipher", "multi"])
def test_sumcheck_generic(bn, kind):
    """sumcheck/prover_test.go:42-94 (genericTest) on the C oracle."""
    n = 1 << bn
    X = [c.from_ints(range(n)), c.from_ints(range(n))]
    if kind == "cipher":
        gate, ark = c.GATE_CIPHER, c.from_u64(145646)
        qs = c.random_fr_array(bn).reshape(1, bn, 4)
        claims = c.evaluation(gate, ark, qs, c.fr(0), X)
    else:
        gate, ark = c.GATE_IDENTITY, None
        ninst = 10
        qs = np.stack([c.from_ints([(i * j + i) for j in range(bn)]).reshape(bn, 4) for i in range(ninst)])
        claims = np.concatenate([c.evaluation(gate, ark, qs[i:i + 1], c.fr(0), X) for i in range(ninst)])
    claim_test = c.evaluation(gate, ark, qs, claims, X)
    rnd = c.mimc_hash(claims)
    assert (c.eval_univariate(claims, rnd) == claim_test).all()
    proof, chal, final = c.sumcheck_prove(gate, ark, X, qs, claims)
    rc, vchal, expected, recomb = c.sumcheck_verify(claims, proof)
    assert rc == 0
    assert (vchal == chal).all() and (recomb == rnd).all()
    g = o.CipherGate(145646) if kind == "cipher" else o.IdentityGate()
    fin = c.to_ints(final)
    assert g.eval(*fin[1:]) * fin[0] % o.Q == c.to_ints(expected)[0]


def test_golden_gkr_file():
    for e in load("gkr_mimc.json"):
        bn = e["bn"]
        i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
        flat, outs, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
        assert fr_to_hex(flat) == e["flat"]
        assert fr_to_hex(outs) == e["outputs"]
        assert c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp) == 0


def test_golden_gkr_bn10_digest():
    """BASELINE config 1 (bN = 10, gkr/gkr_test.go path) against the committed digest."""
    d = load("gkr_mimc_bn10_digest.json")
    bn = d["bn"]
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    flat, outs, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
    assert flat.shape[0] == d["n_elements"]
    assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == d["sha256_flat"]
    assert hashlib.sha256(outs.astype("<u8").tobytes()).hexdigest() == d["sha256_outputs"]
    assert c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp) == 0


@pytest.mark.parametrize("bn", [0, 1, 2, 4, 7, 11])
def test_gkr_claims_consistent_and_verified(bn):
    """gkr/gkr_test.go:14-78: every claim equals Evaluate(layer table, point); Verify accepts;
    a corrupted proof is rejected."""
    i0, qp = c.random_fr_array(1 << bn), c.random_fr_array(bn)
    flat, outs, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
    assert c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp) == 0
    if bn <= 4:
        circ = o.mimc_circuit()
        ins = [o.random_fr_array(1 << bn)] * 2
        a = o.assign(circ, *ins)
        p = o.gkr_prove(circ, a, o.random_fr_array(bn))
        assert o.gkr_proof_to_vec(p) == c.to_ints(flat)
        for layer in range(len(circ)):
            for j, claim in enumerate(p.claims[layer]):
                assert o.evaluate(a[layer], p.q_primes[layer][j]) == claim
    if bn > 0:
        bad = flat.copy()
        bad[5, 0] ^= np.uint64(1)
        assert c.gkr_verify_mimc(bn, bad, i0, i0, outs, qp) != 0


def test_mimc_circuit_is_mimc():
    # examples/mimc_test.go:19-42: a[93][0] == MimcKeyedPermutation(payload[0], key[0])
    bn = 3
    i0 = c.random_fr_array(1 << bn)
    i1 = c.from_ints([7 * i + 3 for i in range(1 << bn)])
    _, outs, _ = c.gkr_prove_mimc(bn, i0, i1, c.random_fr_array(bn))
    for k in range(1 << bn):
        assert (c.mimc_keyed_permutation(i1[k:k + 1], i0[k:k + 1]) == outs[k]).all()
    circ = o.mimc_circuit()
    for lay in circ:  # TestCircuitForm :44-53
        assert lay.Out == sorted(lay.Out)


@pytest.mark.parametrize("bn", [0, 1, 3, 5])
def test_gmimc_circuit_c_oracle_vs_python_oracle(bn):
    """BASELINE config 5: the build-defined GMiMC (t = 2) circuit.  The C oracle's generic-circuit prover agrees
    with the Python oracle, its verifier accepts, and the outputs are hash.GMimcHasher's compression."""
    n = 1 << bn
    circ = o.gmimc_t2_circuit()
    ins = [o.random_fr_array(n), [(3 * i * i + 7) % o.Q for i in range(n)], [(i * i * i * 5 + 1) % o.Q for i in range(n)],
           [o.mimc_hash([i]) for i in range(n)]]
    a = o.assign(circ, *ins)
    qp = o.random_fr_array(bn)
    want = o.gkr_proof_to_vec(o.gkr_prove(circ, a, qp))
    descs = c.circuit_descs(circ)
    cins = [c.from_ints(x) for x in ins]
    flat, outs, _ = c.gkr_prove_circuit(descs, bn, cins, c.from_ints(qp))
    assert c.to_ints(flat) == want
    assert c.to_ints(outs) == [o.gmimc_update([ins[0][k], ins[1][k]], [ins[2][k], ins[3][k]])[0] for k in range(n)]
    assert c.gkr_verify_circuit(descs, bn, flat, cins, outs, c.from_ints(qp)) == 0
    if bn:
        bad = flat.copy()
        bad[3, 0] ^= np.uint64(1)
        assert c.gkr_verify_circuit(descs, bn, bad, cins, outs, c.from_ints(qp)) != 0


@pytest.mark.parametrize("t,nb,bn", [(2, 2, 1), (4, 2, 2)])
def test_gmimc_sponge_circuit_c_oracle_vs_python_oracle(t, nb, bn):
    """The sponge hash.GMimcHasher.Hash over nb blocks as one circuit (layers with two consumers, one-input gates in the
    first block): both oracles produce the same transcript, the restated gkr.Verify accepts it, and the outputs are the
    reference hasher's digests."""
    n = 1 << bn
    circ, imap = o.gmimc_hash_circuit(t, nb)
    msg = [[o.mimc_hash([7 * j + k + 1]) for k in range(n)] for j in range(t * nb)]
    ins = [msg[j] for j in imap]
    a = o.assign(circ, *ins)
    assert a[-1] == [o.gmimc_hash([m[k] for m in msg], t) for k in range(n)]
    qp = o.random_fr_array(bn)
    want = o.gkr_proof_to_vec(o.gkr_prove(circ, a, qp))
    descs = c.circuit_descs(circ)
    cins = [c.from_ints(x) for x in ins]
    flat, outs, _ = c.gkr_prove_circuit(descs, bn, cins, c.from_ints(qp))
    assert c.to_ints(flat) == want and c.to_ints(outs) == a[-1]
    assert c.gkr_verify_circuit(descs, bn, flat, cins, outs, c.from_ints(qp)) == 0


def test_survey_appendix_b():
    """values of an independent restatement of the reference (written by a different
    session, KAT-anchored and verifier-accepted) -- the only second reading of the algorithm this repository
    has, asserted here on BOTH oracles.  Inputs follow sumcheck/testing.go:11-57 and gkr/gkr_test.go:23-25."""
    Q = o.Q
    # common.RandomFrArray(4); poly.FoldedEqTable([2,3]); fold([0,1,2,3], 5)
    rfa4 = [16792413999679, 16792413999678, 16792413999675, 16792413999670]
    assert o.random_fr_array(4) == rfa4 and c.to_ints(c.random_fr_array(4)) == rfa4
    eq23 = [2, (-3) % Q, (-4) % Q, 6]
    assert o.folded_eq_table([2, 3]) == eq23 and c.to_ints(c.folded_eq_table(c.from_ints([2, 3]))) == eq23
    assert o.fold([0, 1, 2, 3], 5) == [10, 11]

    # InitializeCipherGateInstance(2)  (sumcheck/testing.go:11-26)
    X, claims, qs, gate = o.initialize_cipher_gate_instance(2)
    assert qs == [[16792413999679, 16792413999678]]
    assert claims == [6210164148622235469085091026862563128838641017496921984]
    round0 = [21888242871839275184561835838863618971897026818209516123207594011953420612865,
              75362930075014142017900484743430457352086602784186311477888,
              12419049195478414334940816557593184013274926150706173184,
              852695088970139060850731351704017254582492295811584,
              31224423234359380840366957119592272582448808960,
              643155849042919484885696612437680723589120,
              7065388620329840615834319386742878208,
              32340315080560337413961102483456,
              550253821941465088]
    chal = [7683915157646553957972751715506210273774041546390922002351511344957472092635,
            10056654415683796759942491034232422885846793423765035913409411941319351042801]
    final = [12560110031309288988337635978213296499715689218361145289462178438943874440404,
             3536241859137629453641588719987568344846512116130845574414230444658486732454,
             3536241859137629453641588719987568344846512116130845574414230444658486732454]
    proof, ch, fin = o.sumcheck_prove([list(x) for x in X], qs, claims, gate)
    assert proof[0] == round0 and ch == chal and fin == final
    cp, cc, cf = c.sumcheck_prove(c.GATE_CIPHER, c.from_u64(145646), [c.from_ints(x) for x in X],
                                  c.from_ints(qs[0]).reshape(1, 2, 4), c.from_ints(claims))
    assert c.to_ints(cp[0]) == round0 and c.to_ints(cc) == chal and c.to_ints(cf) == final

    # InitializeMultiInstance(2, 3)  (sumcheck/testing.go:28-57)
    X, claims, qs, gate = o.initialize_multi_instance(2, 3)
    assert qs == [[0, 0], [1, 2], [2, 4]] and claims == [0, 4, 8]
    chal = [21366438619878141333539650973972889481318660075116115520302585625232830162408,
            15050161346558720454461840747952731668813985870593406549855952370501809100848]
    fin01 = [6923648215444982917282612078813725970844542082510533916261334677738104788671,
             14006552842636452677048331205383960454354577219993568903064715247815852434430]
    _, ch, fin = o.sumcheck_prove([list(x) for x in X], qs, claims, gate)
    assert ch == chal and fin[:2] == fin01
    _, cc, cf = c.sumcheck_prove(c.GATE_IDENTITY, None, [c.from_ints(x) for x in X],
                                 np.stack([c.from_ints(q) for q in qs]), c.from_ints(claims))
    assert c.to_ints(cc) == chal and c.to_ints(cf)[:2] == fin01

    # gkr.Prove(MimcCircuit), bN = 1, inputs = qPrime = RandomFrArray
    circ = o.mimc_circuit()
    inp = o.random_fr_array(2)
    a = o.assign(circ, inp, list(inp))
    pr = o.gkr_prove(circ, a, o.random_fr_array(1))
    sc93 = [8603423878151217347612283315717020960620926714233984240685120503134163518679,
            6225796057942586806576531527479361841449264046497628497844748636862076955978]
    cl0 = [10987504763916727291808773442712990250720948185337753863556565714056923793942]
    cl1 = [4802317624173645157770282518706233816660915781471186570068121738487279260555]
    assert pr.sumcheck_proofs[93][0][:2] == sc93 and pr.claims[0] == cl0 and pr.claims[1] == cl1
    i0 = c.random_fr_array(2)
    flat, _, _ = c.gkr_prove_mimc(1, i0, i0.copy(), c.random_fr_array(1))
    ints = c.to_ints(flat)
    # flat = GkrProofToVec order: layer 2 (identity, 3 coeffs) then 91 cipher layers of 9 coeffs; claims follow
    assert ints[3 + 90 * 9: 3 + 90 * 9 + 2] == sc93
    assert ints[822] == cl0[0] and ints[823] == cl1[0]
    assert ints == o.gkr_proof_to_vec(pr)
    # proof sizes (coeffs / claims / point coordinates), hints.go:76-116
    for bn, sizes in ((0, (0, 183, 0)), (1, (822, 183, 184)), (3, (2466, 183, 552)), (5, (4110, 183, 920))):
        assert c.mimc_proof_len(bn) == sum(sizes) and o.nb_outputs(circ, bn) == sum(sizes)


@pytest.mark.parametrize("bn", [0, 1, 3])
def test_variadic_gates_c_vs_python(bn):
    """The build-defined variadic gates (sum / sum^7 over 3 and 4 inputs) in a small layered circuit: the two
    restatements produce the same transcript, and the restated gkr.Verify accepts it."""
    n = 1 << bn
    L = [o.Layer([]) for _ in range(4)]
    L.append(o.Layer([0], o.IdentityGate()))                       # 4: copy of input 0 (used twice)
    L.append(o.Layer([4, 1, 2], o.SumGate(o.ARKS[0], 7)))          # 5: (x0 + x1 + x2 + Ark)^7
    L.append(o.Layer([5, 4, 3], o.SumGate(o.ARKS[1], 1)))          # 6: y + x0 + x3 + Ark
    L.append(o.Layer([6], o.IdentityGate()))                       # 7: copy (used twice)
    L.append(o.Layer([7], o.SumGate(5, 7)))                        # 8: one-input power gate
    L.append(o.Layer([7, 8], o.AddGate(o.ARKS[2])))                # 9
    circ = o.build_circuit(L)
    rng = np.random.default_rng(bn)
    ins = [[int(v) for v in rng.integers(0, 1 << 62, n)] for _ in range(4)]
    qp = o.random_fr_array(bn)
    a = o.assign(circ, *ins)
    pr = o.gkr_prove(circ, a, qp)
    assert o.gkr_verify(circ, pr, ins, a[-1], qp)
    descs = c.circuit_descs(circ)
    cf, cout, _ = c.gkr_prove_circuit(descs, bn, [c.from_ints(x) for x in ins], c.from_ints(qp) if bn else c.fr(0))
    assert c.to_ints(cf) == o.gkr_proof_to_vec(pr) and c.to_ints(cout) == a[-1]
    assert c.gkr_verify_circuit(descs, bn, cf, [c.from_ints(x) for x in ins], cout, c.from_ints(qp) if bn else c.fr(0)) == 0


def test_fr_mul_variants_agree():
    """The unrolled no-carry CIOS (fr.Element.Mul as gnark-crypto's assembly does it) and the looped five-word CIOS of
    rounds 1-2 give the same canonical products, corner values included."""
    import numpy as np
    rng = np.random.default_rng(7)
    qm1 = c.from_ints([21888242871839275222246405745257275088548364400416034343698204186575808495616])
    vals = [c.fr(1)[0:1], c.from_u64(1), qm1, c.from_ints([(1 << 253) + 12345]), c.random_fr_array(6)[5:6]]
    vals += [c.from_ints([int.from_bytes(rng.bytes(32), "little") % 21888242871839275222246405745257275088548364400416034343698204186575808495617])
             for _ in range(40)]
    for a in vals:
        for b in vals:
            o1, o2 = c.fr(1), c.fr(1)
            c.lib.oracle_fr_mul(o1.ctypes.data, a.ctypes.data, b.ctypes.data)
            c.lib.oracle_fr_mul_generic(o2.ctypes.data, a.ctypes.data, b.ctypes.data)
            assert np.array_equal(o1, o2)


def test_compute_h_restatement_against_schoolbook_division():
    """oracle/pyoracle_fft.py (computeH as prover/gadget/prove.go:308-359 runs it on gnark-crypto's fft.Domain -- an
    un-vendored dependency, so "parity unpinned") against mathematics: for satisfied constraints (c = a*b on the domain)
    the result is the quotient (A*B - C) / (X^n - 1) by schoolbook polynomial arithmetic, at the bit-reversed positions;
    for arbitrary c it agrees with (A*B - C) * (-2)^-1 on the odd coset; the committed fixtures are what it produces."""
    import json
    import random
    import pyoracle_fft as F
    random.seed(5)
    for n in (1, 2, 4, 8, 16, 32):
        dom = F.Domain(n, 1)
        assert pow(dom.generator, n, F.Q) == 1 and (n == 1 or pow(dom.generator, n // 2, F.Q) == F.Q - 1)
        assert pow(dom.finer_generator, n, F.Q) == F.Q - 1                 # Z = X^n - 1 is -2 on the coset
        a = [random.randrange(F.Q) for _ in range(n)]
        b = [random.randrange(F.Q) for _ in range(n)]
        c = [x * y % F.Q for x, y in zip(a, b)]
        H = F.h_by_division(a, b, c, dom)
        assert F.compute_h(a, b, c) == [H[F.bit_reverse(p, dom.log)] for p in range(n)]
    n = 16
    dom = F.Domain(n, 1)
    a, b, c = ([random.randrange(F.Q) for _ in range(n)] for _ in range(3))
    h = F.compute_h(a, b, c)
    Hc = [h[F.bit_reverse(k, dom.log)] for k in range(n)]
    A, B, C = (F.interpolate_naive(v, dom) for v in (a, b, c))
    ev = lambda P, x: sum(co * pow(x, k, F.Q) for k, co in enumerate(P)) % F.Q      # noqa: E731
    for j in range(n):
        x = dom.finer_generator * pow(dom.generator, j, F.Q) % F.Q
        assert ev(Hc, x) * (F.Q - 2) % F.Q == (ev(A, x) * ev(B, x) - ev(C, x)) % F.Q
    # FFT / FFTInverse round trips in both decimations and on the coset
    v = [random.randrange(F.Q) for _ in range(n)]
    w = list(v)
    F.fft(dom, w, "DIF", 0)
    assert [w[F.bit_reverse(i, dom.log)] for i in range(n)] == [ev(v, pow(dom.generator, i, F.Q)) for i in range(n)]
    F.fft_inverse(dom, w, "DIT", 0)
    assert w == v
    F.fft(dom, w, "DIF", 1)
    F.fft_inverse(dom, w, "DIT", 1)
    assert w == v
    for e in load("compute_h.json"):
        got = F.compute_h([int(x, 16) for x in e["a"]], [int(x, 16) for x in e["b"]], [int(x, 16) for x in e["c"]], e["cardinality"] or None)
        assert [hex(x) for x in got] == e["h"]


def test_round_sums_are_degree_7_in_the_previous_challenge():
    """The fact the speculative rounds rest on (cipher_spec.hip.h): with the tables of round k being the tables of round
    k-1 folded with r (poly/multilin.go:27-34), the monomial sums M_j(r) = sum_x W(x) u_x(r)^(7-j) d_x(r)^j of the cipher
    gate's round are polynomials of degree 7 in r -- so their values at r = 0..7 determine them, and Lagrange interpolation
    on those eight points reproduces the sums at any challenge exactly.  Plain Python integers, no library code."""
    import random
    q = o.Q
    rng = random.Random(7)
    P = 4                                                     # pairs of round k; the previous round's tables have 4P entries
    K = [rng.randrange(q) for _ in range(4 * P)]
    S = [rng.randrange(q) for _ in range(4 * P)]
    W = [rng.randrange(q) for _ in range(P)]
    ark = rng.randrange(q)

    def sums(r):
        fold = lambda t: [(t[i] + r * (t[i + 2 * P] - t[i])) % q for i in range(2 * P)]     # binds the top index bit
        k, s = fold(K), fold(S)
        out = []
        for j in range(8):
            acc = 0
            for x in range(P):
                u = (k[x] + s[x] + ark) % q
                d = ((k[x + P] - k[x]) + (s[x + P] - s[x])) % q
                acc += W[x] * pow(u, 7 - j, q) * pow(d, j, q)
            out.append(acc % q)
        return out

    cand = [sums(i) for i in range(8)]
    for r in (8, 12345, rng.randrange(q), q - 1, 3):          # (3: one of the points themselves)
        L = []
        for i in range(8):
            num = den = 1
            for jj in range(8):
                if jj != i:
                    num = num * (r - jj) % q
                    den = den * (i - jj) % q
            L.append(num * pow(den, q - 2, q) % q)
        got = [sum(L[i] * cand[i][j] for i in range(8)) % q for j in range(8)]
        assert got == sums(r), r
