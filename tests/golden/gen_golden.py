#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the Python big-int restatement
(oracle/pyoracle.py).  Run in the development container: `python tests/golden/gen_golden.py`.

The reference is pure Go and cannot be executed here, so these vectors are NOT Go outputs; they are
anchored on the reference's known-answer tests (asserted below before anything is written) and on
verifier acceptance.  Elements are stored as 64-hex-digit strings: the four little-endian u64
Montgomery limbs of gnark-crypto's fr.Element, limb 0 first (pyoracle.to_hex)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle as o  # noqa: E402


def H(xs):
    return [o.to_hex(x) for x in xs]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", name)


def main():
    # --- anchors: the reference's own known answers -------------------------------------------------
    assert o.mimc_hash([12]) == 1808205620575546259657963589762746470347087906694759866517376279978241663265  # hash/hash_test.go:21-27
    assert o.fold([0, 1, 2, 3], 5) == [10, 11]                                  # poly/multilin_test.go:12-31
    L = o.lagrange_coefficient(7)[2]
    assert [o.eval_univariate(L, i) for i in range(7)] == [0, 0, 1, 0, 0, 0, 0]  # poly/lagrange_test.go:10-29
    assert (o.eval_univariate([4, 3, 2, 1], 0) + o.eval_univariate([4, 3, 2, 1], 1)) % o.Q == 14  # snark/polynomial/univariate_test.go:39-45
    assert o.eval_univariate([4, 3, 2, 1], 5) == 194

    kat = {
        "q": str(o.Q),
        "mimc_hash": [{"in": H(v), "out": o.to_hex(o.mimc_hash(v))} for v in
                      [[12], [0], [1, 2, 3], o.random_fr_array(9), list(range(91))]],
        "mimc_hash_12_decimal": str(o.mimc_hash([12])),
        "mimc_perm": [{"x": o.to_hex(x), "key": o.to_hex(k), "out": o.to_hex(o.mimc_keyed_permutation(x, k))}
                      for x, k in [(0, 0), (1, 2), (o.Q - 1, 12345), tuple(o.random_fr_array(7)[5:7])]],
        "random_fr_array_16": H(o.random_fr_array(16)),
        "mont_12": o.to_hex(12),
        "gmimc_hash": [{"t": t, "in": H(m), "out": o.to_hex(o.gmimc_hash(m, t))}
                       for t in (2, 4, 8) for m in ([1, 2, 3, 4, 5, 6, 7, 8, 9], [12])],
        "lagrange_9": [H(r) for r in o.lagrange_coefficient(9)],
        "lagrange_3": [H(r) for r in o.lagrange_coefficient(3)],
    }
    dump("kat.json", kat)

    poly = {"fold": [], "eq": [], "eval_eq": []}
    for bn in range(1, 7):
        t = o.random_fr_array(1 << bn)
        r = o.random_fr_array(bn + 3)[-1]
        poly["fold"].append({"tbl": H(t), "r": o.to_hex(r), "out": H(o.fold(t, r))})
    poly["fold"].append({"tbl": H([0, 1, 2, 3]), "r": o.to_hex(5), "out": H([10, 11])})
    for bn in range(0, 7):
        q = o.random_fr_array(bn)
        m = o.mimc_hash([bn])
        poly["eq"].append({"q": H(q), "mult": None, "out": H(o.folded_eq_table(q))})
        poly["eq"].append({"q": H(q), "mult": o.to_hex(m), "out": H(o.folded_eq_table(q, m))})
        h = [o.mimc_hash([i, bn]) for i in range(bn)]
        poly["eval_eq"].append({"q": H(q), "h": H(h), "out": o.to_hex(o.eval_eq(q, h))})
    dump("poly.json", poly)

    sc = []
    for bn in range(0, 7):
        X, claims, qs, gate = o.initialize_cipher_gate_instance(bn)
        proof, ch, fc = o.sumcheck_prove(X, qs, claims, gate)
        sc.append({"kind": "cipher", "ark": o.to_hex(145646), "bn": bn, "claims": H(claims),
                   "qprimes": [H(q) for q in qs], "proof": [H(r) for r in proof], "challenges": H(ch),
                   "final": H(fc)})
        for ninst in (3, 10):
            X, claims, qs, gate = o.initialize_multi_instance(bn, ninst)
            proof, ch, fc = o.sumcheck_prove(X, qs, claims, gate)
            sc.append({"kind": "identity", "ark": None, "bn": bn, "ninstance": ninst, "claims": H(claims),
                       "qprimes": [H(q) for q in qs], "proof": [H(r) for r in proof], "challenges": H(ch),
                       "final": H(fc)})
    dump("sumcheck.json", sc)

    circ = o.mimc_circuit()
    gk = []
    for bn in (0, 1, 2, 3, 5):
        ins = [o.random_fr_array(1 << bn), o.random_fr_array(1 << bn)]
        qp = o.random_fr_array(bn)
        a = o.assign(circ, *ins)
        p = o.gkr_prove(circ, a, qp)
        assert o.gkr_verify(circ, p, ins, a[93], qp)
        v = o.gkr_proof_to_vec(p)
        assert len(v) == o.nb_outputs(circ, bn)
        gk.append({"bn": bn, "flat": H(v), "outputs": H(a[93])})
    dump("gkr_mimc.json", gk)

    # BASELINE config 1: bN = 10 -- digest only (SHA-256 over the limb stream, little-endian u64s)
    bn = 10
    ins = [o.random_fr_array(1 << bn), o.random_fr_array(1 << bn)]
    qp = o.random_fr_array(bn)
    a = o.assign(circ, *ins)
    p = o.gkr_prove(circ, a, qp)
    assert o.gkr_verify(circ, p, ins, a[93], qp)
    v = o.gkr_proof_to_vec(p)
    def digest(vals):
        h = hashlib.sha256()
        for x in vals:
            for l in o.to_mont_limbs(x):
                h.update(int(l).to_bytes(8, "little"))
        return h.hexdigest()
    dump("gkr_mimc_bn10_digest.json", {"bn": bn, "n_elements": len(v), "sha256_flat": digest(v),
                                       "sha256_outputs": digest(a[93]),
                                       "first": H(v[:4]), "last": H(v[-4:])})


if __name__ == "__main__":
    main()
