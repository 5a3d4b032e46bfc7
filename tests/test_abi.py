"""CPU tests of the boundary: the C-ABI library loads without a GPU and exports every symbol that
include/gkrhip.h declares; the generated Montgomery schedule (portable branch) matches the oracle;
the product never references oracle/."""
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gk():
    b = importlib.import_module("gkr-mimc_amd.build")
    b.build()
    return importlib.import_module("gkr-mimc_amd")


def test_library_exports_every_declared_symbol(gk):
    hdr = open(os.path.join(ROOT, "include", "gkrhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gkrhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = gk.prover.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(gk.prover.ABI), declared ^ set(gk.prover.ABI)


def test_no_gpu_fails_loudly(gk):
    if gk.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gk.prover.GkrHipError):
        gk.init(0)
    import numpy as np
    with pytest.raises(gk.prover.GkrHipError):
        gk.fold(np.zeros((4, 4), np.uint64), np.zeros((1, 4), np.uint64))


def test_sumcheck_verify_entry_point_vs_oracle(gk):
    """gkrhip_sumcheck_verify (sumcheck/verifier.go:28-56, host-only) against the oracle's restated verifier on oracle-made
    proofs: cipher gate with one claim, identity gate with several claims, every output, and corrupted proofs (the round
    of the rejection and the reference's error text)."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as c
    for bn, gate, ninst in ((0, c.GATE_CIPHER, 1), (1, c.GATE_CIPHER, 1), (6, c.GATE_CIPHER, 1), (5, c.GATE_IDENTITY, 3), (7, c.GATE_IDENTITY, 10)):
        n = 1 << bn
        X = [c.random_fr_array(n), c.from_ints([(7 * i * i + 3) % 1000003 for i in range(n)])]
        ark = c.from_u64(145646)
        qs = np.stack([c.from_ints([(i * j + i + 1) for j in range(bn)]) if bn else c.fr(0) for i in range(ninst)]).reshape(ninst, bn, 4)
        claims = np.concatenate([c.evaluation(gate, ark, qs[i:i + 1], c.fr(0), X) for i in range(ninst)])
        proof, chal, _fin = c.sumcheck_prove(gate, ark, X, qs, claims)
        rc, ochal, ofinal, orecomb = c.sumcheck_verify(claims, proof)
        assert rc == 0
        gchal, gfinal, grecomb = gk.sumcheck_verify(claims, proof)
        assert np.array_equal(gchal, ochal) and np.array_equal(gchal, chal)
        assert np.array_equal(gfinal, ofinal) and np.array_equal(grecomb, orecomb)
        for rnd in range(bn):
            bad = proof.copy()
            bad[rnd, rnd % proof.shape[1], 1] ^= np.uint64(4)
            assert c.sumcheck_verify(claims, bad)[0] != 0
            with pytest.raises(gk.prover.GkrHipError, match=r"at round %d verifier eval at 0 \+ 1 = \d+ \|\| expected = \d+" % rnd):
                gk.sumcheck_verify(claims, bad)
    with pytest.raises(gk.prover.GkrHipError, match="no claim"):
        gk.sumcheck_verify(np.zeros((0, 4), np.uint64), np.zeros((1, 3, 4), np.uint64))


def test_error_codes_are_readable_from_another_thread(gk, tmp_path):
    """tests/cpp/test_abi_errors.cpp: every failure has a code of its own and gkrhip_last_error_r(code) answers from any
    thread (a goroutine may have migrated between the failing cgo call and must())."""
    exe = str(tmp_path / "test_abi_errors")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_abi_errors.cpp"), "-ldl", "-pthread"])
    out = subprocess.run([exe, gk.prover.library_path()], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ABI-ERRORS-OK" in out.stdout, out.stdout + out.stderr
    import numpy as np
    with pytest.raises(gk.prover.GkrHipError) as e1:
        gk.sumcheck_verify(np.zeros((0, 4), np.uint64), np.zeros((1, 3, 4), np.uint64))
    assert "no claim" in str(e1.value)


def test_gmimc_circuit_description_matches_oracle(gk):
    """The library's build-defined GMiMC (t = 2) circuit (BASELINE config 5) is layer for layer the circuit the
    Python oracle proves, and that circuit computes hash.GMimcHasher's compression (checked in pyoracle)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as o
    lib_c = gk.gmimc_t2_circuit()
    ref_c = o.gmimc_t2_circuit()
    assert len(lib_c) == len(ref_c) == 100
    kinds = {None: -1, "identity": gk.GATE_IDENTITY, "cipher": gk.GATE_CIPHER, "add": gk.GATE_ADD}
    for (gate, ins, ark), lay in zip(lib_c, ref_c):
        assert gate == kinds[lay.gate.kind if lay.gate else None]
        assert ins == lay.In
        if gate in (gk.GATE_CIPHER, gk.GATE_ADD):
            assert ark == o.to_mont_limbs(lay.gate.ark)
    # the circuit is the compression function of the reference's hasher
    ins = [[5], [7], [11], [13]]
    a = o.assign(ref_c, *ins)
    assert a[-1][0] == o.gmimc_update([5, 7], [11, 13])[0]


def test_gmimc_t_circuits_match_oracle(gk):
    """gkrhip_gmimc_circuit(t) for t = 2, 4, 8 (hash/gmimc.go:16-20) is layer for layer the circuit pyoracle builds
    independently -- the last layer being the registered three-input gate -- with the same input map, and that circuit
    computes hash.GMimcHasher's compression on every instance."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as o
    for t in (2, 4, 8):
        lib_c, lib_map = gk.gmimc_circuit(t)
        ref_c, ref_map = o.gmimc_circuit(t)
        assert lib_map == ref_map and len(lib_c) == len(ref_c)
        for (gate, ins, ark), lay in zip(lib_c, ref_c):
            assert ins == lay.In
            if lay.gate is None:
                assert gate == -1
                continue
            if lay.gate.kind == "sum":
                d = gk.gate_lookup(gate)
                assert (d["n_in"], d["sum_mask"], d["power"]) == (len(lay.In), (1 << len(lay.In)) - 1, 1) and gate >= 3
            else:
                assert gate == {"identity": gk.GATE_IDENTITY, "cipher": gk.GATE_CIPHER, "add": gk.GATE_ADD}[lay.gate.kind]
            if lay.gate.kind != "identity":
                assert ark == o.to_mont_limbs(lay.gate.ark)
        random.seed(t)
        state = [[random.randrange(o.Q) for _ in range(3)] for _ in range(t)]
        block = [[random.randrange(o.Q) for _ in range(3)] for _ in range(t)]
        a = o.assign(ref_c, *[(state[j] if j < t else block[j - t]) for j in ref_map])
        for k in range(3):
            assert a[-1][k] == o.gmimc_update([s[k] for s in state], [b[k] for b in block])[0]
    # and the compression is the one the committed hash vectors were produced with (kat.json gmimc_hash, all t)
    from util import hex_to_fr, load
    import coracle as c
    for e in load("kat.json")["gmimc_hash"]:
        msg = c.to_ints(hex_to_fr(e["in"]))
        assert o.gmimc_hash(msg, e["t"]) == c.to_ints(hex_to_fr(e["out"]))[0]


def test_gmimc_hash_circuits_match_oracle(gk):
    """gkrhip_gmimc_hash_circuit(t, nblocks) -- the whole sponge hash.GMimcHasher.Hash (hash/gmimc.go:29-49) over nblocks
    blocks -- is layer for layer the circuit pyoracle builds, with the same input map, and that circuit's output is the
    reference hasher's digest of each instance's message (the hasher itself is pinned on kat.json's gmimc_hash vectors)."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as o
    for t, nb in ((2, 1), (2, 2), (2, 3), (4, 2), (8, 2)):
        lib_c, lib_map = gk.gmimc_hash_circuit(t, nb)
        ref_c, ref_map = o.gmimc_hash_circuit(t, nb)
        assert lib_map == ref_map and len(lib_c) == len(ref_c)
        for (gate, ins, ark), lay in zip(lib_c, ref_c):
            assert ins == lay.In
            if lay.gate is None:
                assert gate == -1
                continue
            if lay.gate.kind in ("sum", "sum_pow7"):
                d = gk.gate_lookup(gate)
                assert (d["n_in"], d["sum_mask"], d["power"]) == (len(lay.In), (1 << len(lay.In)) - 1, lay.gate.power) and gate >= 3
            else:
                assert gate == {"identity": gk.GATE_IDENTITY, "cipher": gk.GATE_CIPHER, "add": gk.GATE_ADD}[lay.gate.kind]
            if lay.gate.kind != "identity":
                assert ark == o.to_mont_limbs(lay.gate.ark)
        random.seed(10 * t + nb)
        msg = [[random.randrange(o.Q) for _ in range(2)] for _ in range(t * nb)]
        a = o.assign(ref_c, *[msg[k] for k in ref_map])
        for k in range(2):
            assert a[-1][k] == o.gmimc_hash([m[k] for m in msg], t)
    with pytest.raises(gk.prover.GkrHipError):
        gk.gmimc_hash_circuit(3, 2)
    with pytest.raises(gk.prover.GkrHipError):
        gk.gmimc_hash_circuit(2, 0)


def test_gate_table(gk):
    """The gate plug point (circuit/gates.go:9-21) as a descriptor table: built-ins, registration, refusals."""
    assert gk.gate_lookup(gk.GATE_IDENTITY) == {"id": "CopyGate", "n_in": 1, "sum_mask": 1, "power": 1}
    assert gk.gate_lookup(gk.GATE_CIPHER)["power"] == 7 and gk.gate_lookup(gk.GATE_CIPHER)["n_in"] == 2
    g = gk.gate_register("test-sum4-pow7", 4, 0b1111, 7)
    assert g >= 3 and gk.gate_register("test-sum4-pow7", 4, 0b1111, 7) == g          # idempotent
    assert gk.gate_lookup(g) == {"id": "test-sum4-pow7", "n_in": 4, "sum_mask": 15, "power": 7}
    assert gk.gate_degree(g) == 7
    for bad in (("test-mul", 2, 3, 2), ("test-none", 2, 0, 1), ("test-wide", 5, 31, 1), ("test-mask", 2, 4, 1),
                ("test-sum4-pow7", 3, 7, 1)):          # power 2; empty sum; 5 inputs; mask outside the inputs; ID reused
        with pytest.raises(gk.prover.GkrHipError):
            gk.gate_register(*bad)
    with pytest.raises(gk.prover.GkrHipError):
        gk.gate_lookup(10 ** 6)


def test_proof_len(gk):
    for bn in (0, 1, 5, 24):
        assert gk.mimc_proof_len(bn) == 822 * bn + 183 + 184 * bn


def test_montgomery_schedule_portable_branch(tmp_path):
    """fr_bn254.h + fr_mont_gen.inc (host branch, identical column schedule to the device asm) vs oracle."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    exe = str(tmp_path / "test_fr")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_fr_schedule.cpp"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lgkr_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-fopenmp"])
    out = subprocess.check_output([exe]).decode()
    assert "bad=0" in out, out


def test_host_scalar_code_vs_oracle(tmp_path):
    """fr_host.h (Fiat-Shamir MiMC hash written for latency, interpolation, limb-split reduction) vs oracle: the portable
    source, and -- as the product is built, with BMI2 and ADX -- the hash's products as mulx/adcx/adox assembly."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    flags = open("/proc/cpuinfo").read()
    variants = [[]] + ([["-mbmi2", "-madx"]] if " adx" in flags and " bmi2" in flags else [])
    for k, extra in enumerate(variants):
        exe = str(tmp_path / ("test_host%d" % k))
        subprocess.check_call(["g++", "-O2"] + extra + ["-o", exe, os.path.join(ROOT, "tests", "cpp", "test_host_fr.cpp"),
                               "-L" + os.path.join(ROOT, "oracle"), "-lgkr_oracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-fopenmp"])
        out = subprocess.check_output([exe]).decode()
        assert "bad=0" in out, (extra, out)


def test_generated_schedule_is_current():
    """Every generated column schedule (product, square, interleaved pair, wide MAC, constant-multiplier product) is what
    tools/gen_mont_asm.py produces now."""
    incs = [os.path.join(ROOT, "gkr-mimc_amd", "csrc", f) for f in
            ("fr_mont_gen.inc", "fr_sqr_gen.inc", "fr_mont2_gen.inc", "fr_mac_wide_gen.inc", "fr_mulc2_gen.inc")]
    before = [open(f).read() for f in incs]
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_mont_asm.py")], stdout=subprocess.DEVNULL)
    assert [open(f).read() for f in incs] == before


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "gkr-mimc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                for bad in ("gkr_oracle", "coracle", "pyoracle", "libgkr_oracle"):
                    assert bad not in txt, (f, bad)


def test_header_is_plain_c():
    """include/gkrhip.h is what a cgo preamble includes: it must be valid C99 on its own."""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "gkrhip.h")])


# ---------------------------------------------------------------- round 5: the prover's own check and round 0 ahead, host parts
def test_host_sumcheck_closes_on_oracle_proofs(gk):
    """gkrhip_host_sumcheck_closes (what gkr.Prove runs on every sumcheck before returning it) accepts the oracle's sumchecks --
    cipher gate at one point, identity gate at five points -- and names what is wrong with a corrupted one: a round, the
    closing identity, finalClaims[0]; a claim that is not the sum fails round 0 only when the caller vouches for the claims."""
    import numpy as np
    import coracle as c
    bn = 6
    n = 1 << bn
    rng = np.random.default_rng(5)
    X = [c.random_fr_array(n), c.from_ints([int(v) for v in rng.integers(0, 1 << 62, n)])]
    ark = c.from_u64(145646)
    qs = c.random_fr_array(bn).reshape(1, bn, 4)
    claims = c.evaluation(c.GATE_CIPHER, ark, qs, c.fr(0), X)
    proof, chal, fin = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims)
    args = (gk.GATE_CIPHER, ark, 2, qs, claims)
    assert gk.host_sumcheck_closes(*args, proof, chal, fin) == 0
    for (i, j), want in (((0, 0), 1), ((3, 4), 4), ((5, 8), 6)):      # a coefficient of round i breaks round i's check (or the next one's)
        bad = proof.copy()
        bad[i, j, 0] ^= np.uint64(1)
        assert gk.host_sumcheck_closes(*args, bad, chal, fin) in (want, want + 1, -1), (i, j)
    bad = fin.copy(); bad[1, 0] ^= np.uint64(2)
    assert gk.host_sumcheck_closes(*args, proof, chal, bad) == -1
    bad = fin.copy(); bad[0, 1] ^= np.uint64(2)
    assert gk.host_sumcheck_closes(*args, proof, chal, bad) == -2
    bad = chal.copy(); bad[bn - 1, 0] ^= np.uint64(1)
    assert gk.host_sumcheck_closes(*args, proof, bad, fin) != 0
    wrong = claims.copy(); wrong[0, 0] ^= np.uint64(1)
    assert gk.host_sumcheck_closes(gk.GATE_CIPHER, ark, 2, qs, wrong, proof, chal, fin) == 1
    assert gk.host_sumcheck_closes(gk.GATE_CIPHER, ark, 2, qs, wrong, proof, chal, fin, claims_are_sums=False) == 0
    # the output layer of gkr.Prove: no claim at all (sumcheck/prover.go:121), the chain starts at P_0(r_0)
    p0, c0, f0 = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, c.fr(0))
    assert gk.host_sumcheck_closes(gk.GATE_CIPHER, ark, 2, qs, c.fr(0), p0, c0, f0) == 0
    # identity gate, five claims on one table: the recombination challenge and the eq values weighted by its powers
    ninst = 5
    Xi = [c.random_fr_array(n)]
    qm = np.stack([c.from_ints([(i * j + i + 1) for j in range(bn)]).reshape(bn, 4) for i in range(ninst)])
    cl = np.concatenate([c.evaluation(c.GATE_IDENTITY, None, qm[i:i + 1], c.fr(0), Xi) for i in range(ninst)])
    pm, cm, fm = c.sumcheck_prove(c.GATE_IDENTITY, None, Xi, qm, cl)
    assert gk.host_sumcheck_closes(gk.GATE_IDENTITY, None, 1, qm, cl, pm, cm, fm) == 0
    bad = cl.copy(); bad[3, 2] ^= np.uint64(1)
    assert gk.host_sumcheck_closes(gk.GATE_IDENTITY, None, 1, qm, bad, pm, cm, fm) != 0
    bad = pm.copy(); bad[2, 1, 0] ^= np.uint64(4)
    assert gk.host_sumcheck_closes(gk.GATE_IDENTITY, None, 1, qm, cl, bad, cm, fm) in (3, 4)


def test_round0_ahead_factorisation(gk):
    """The identity round 0 ahead of its point rests on, in Python integers: with the pair index x = (x_hi, y),
    y the low t bits, sum_x eq(q[1:], x) m(x) = sum_y eq(q[m-t:], y) S(y), S(y) = sum_{x_hi} eq(q[1:m-t], x_hi) m(x_hi, y) -- and
    the library's host-side contraction (gkrhip_host_ahead_contract) of seven such class-sum tables."""
    import numpy as np
    import coracle as c
    import pyoracle as o
    Q = o.Q
    m, t = 7, 3
    P = 1 << (m - 1)
    q = [int(v) for v in o.random_fr_array(m)]
    rng = np.random.default_rng(11)
    ms = [[int(v) % Q for v in rng.integers(1, 1 << 62, P)] for _ in range(7)]        # seven monomial tables m_j(x)
    full = o.folded_eq_table(q[1:])                              # eq(q[1:], x), q[1] <-> the most significant bit of x
    hi = o.folded_eq_table(q[1:m - t])
    lo = o.folded_eq_table(q[m - t:])
    want = [sum(full[x] * mj[x] for x in range(P)) % Q for mj in ms]
    S = [[sum(hi[xh] * mj[(xh << t) | y] for xh in range(1 << (m - 1 - t))) % Q for y in range(1 << t)] for mj in ms]
    assert [sum(lo[y] * Sj[y] for y in range(1 << t)) % Q for Sj in S] == want
    flat = c.from_ints([v for Sj in S for v in Sj])
    got = gk.host_ahead_contract(flat, c.from_ints(q[m - t:]))
    assert c.to_ints(got) == want


def test_proof_group_driver_without_a_gpu(gk):
    """host_group.hip.h on the host alone (gkrhip_host_group_selftest): proofs on stacks of their own ask for launches through
    launch_batch, a recorder stands in for hipLaunchKernel.  Launches that agree go out as ONE with the arguments of every proof in it,
    in each proof's order; a proof that asks for another grid gets a launch of its own at that step and is back in the common launch
    at the next; a proof that returns early leaves the group without holding the others up."""
    for n in (1, 2, 3, 8):
        wanted, made, most, verdict = gk.host_group_selftest(n, 50)
        assert verdict == 0 and wanted == 50 * n and made == 50 and most == n, (n, wanted, made, most, verdict)
    # proof 1 parts ways at step 7: that step takes two launches (its own and the others'), every other step one
    wanted, made, most, verdict = gk.host_group_selftest(4, 40, 7, -1)
    assert verdict == 0 and wanted == 160 and made == 41 and most == 4, (wanted, made, most, verdict)
    # the last of five proofs returns after 10 of 30 steps: 10 launches of five, then 20 of four
    wanted, made, most, verdict = gk.host_group_selftest(5, 30, -1, 10)
    assert verdict == 0 and wanted == 4 * 30 + 10 and made == 30 and most == 5, (wanted, made, most, verdict)
    # both at once, and a proof that asks for nothing at all
    wanted, made, most, verdict = gk.host_group_selftest(3, 25, 24, 0)
    assert verdict == 0 and wanted == 50 and made == 26 and most == 2, (wanted, made, most, verdict)
    with pytest.raises(gk.GkrHipError):
        gk.host_group_selftest(9, 5)
