"""GPU tests (-m gpu) of the prover's own integrity checks (host_sumcheck.hip.h: sumcheck_closes / sumcheck_prove_dev).

Inside gkr.Prove the fused round loops derive one monomial sum of every round from the running claim, so a slip of the
device side -- a race, incomplete look-ahead products, a bad fold -- used to produce rounds that agree with each other and
a transcript that is simply wrong, with rc 0.  Now every sumcheck is held against the verifier's identities
(sumcheck/verifier.go:41-47, gkr/verifier.go:93-114) before it is returned and run once more in safe mode if it does not
close.  The slips are provoked with the fault-injection options of gkrhip_set_option (never the environment):
test_corrupt_sum = k flips one bit of a device sum of round k, test_corrupt_tail flips one bit of the table entries a
fused loop hands to the host.  Every case: the oracle's transcript, counter layer_check_failures == 1."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = textwrap.dedent("""
    import importlib, sys, threading
    import numpy as np
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import coracle as c
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    import os
    for kv in filter(None, os.environ.get("GKR_CASE_OPTIONS", "").split(",")):      # library options (not environment switches)
        gk.set_option(kv.split("=")[0], int(kv.split("=")[1]))
""") % (ROOT, os.path.join(ROOT, "oracle"))


def _run(body, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "INTEGRITY-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


SOLO = """
    bn = 12
    s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
    qp = c.random_fr_array(bn)
    i0 = c.random_fr_array(1 << bn)
    want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
    gk.profile_reset(0)
    assert np.array_equal(s.prove(qp), want)
    per = gk.profile_get()["layer_checks"]
    assert per == 92 and gk.profile_get()["layer_check_failures"] == 0, gk.profile_get()       # 91 cipher layers + the key-copy layer
    for k in %s:
        for layer in (0, 1, 47, 90):      # sumchecks in the order they are proven: the output layer (no claim), ..., the first cipher layer
            gk.profile_reset(0)
            gk.set_option("test_corrupt_sum", k)
            gk.set_option("test_corrupt_skip", layer)
            for _ in range(2):
                assert np.array_equal(s.prove(qp), want), ("round", k, layer)
            p = gk.profile_get()
            assert p["layer_check_failures"] == 1 and p["layer_checks"] == 2 * per, (k, layer, p)
    # the key-copy layer: 91 claims, 91 points, the reference-shaped rounds with a device-built Eq table
    gk.profile_reset(0)
    gk.set_option("test_corrupt_sum", 2)
    gk.set_option("test_corrupt_skip", 91)
    assert np.array_equal(s.prove(qp), want), "copy layer"
    assert gk.profile_get()["layer_check_failures"] == 1, gk.profile_get()
    gk.profile_reset(0)
    gk.set_option("test_corrupt_tail", 1)
    assert np.array_equal(s.prove(qp), want), "tail"
    assert gk.profile_get()["layer_check_failures"] == 1, gk.profile_get()
    print("INTEGRITY-OK")
"""


def test_corrupted_sum_or_fold_is_caught_and_the_layer_rerun():
    """One proof alone (every serial-latency path at its default), the same with every path forced on or off, the
    reference-shaped evaluator, a small thread budget: a flipped bit in round 0, in a middle round and in the last device
    round, then a flipped bit in the exported tables."""
    _run(SOLO % "(0, 1, 4)")
    _run(SOLO % "(0, 3, 5)", {"GKRHIP_PRELAUNCH": "2", "GKRHIP_SPEC": "2", "GKRHIP_PRE": "2"})
    _run(SOLO % "(0, 2, 5)", {"GKRHIP_PRELAUNCH": "2", "GKRHIP_SPEC": "0", "GKRHIP_COOP": "2"})
    _run(SOLO % "(0, 2, 11)", {"GKRHIP_PRELAUNCH": "0", "GKRHIP_PRE": "0", "GKRHIP_SPEC": "0", "GKRHIP_COOP": "0", "GKRHIP_HOST_TAIL": "0"})
    _run(SOLO % "(0, 3)", {"GKRHIP_GMAX": "8", "GKR_CASE_OPTIONS": "claim_trick=0"})


def test_corrupted_sum_in_the_reference_shaped_rounds():
    """GKRHIP_GENERIC=1: every layer through k_partial_eval + k_fold (no sum is derived there: the round checks see the slip)."""
    _run("""
        bn = 9
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
        qp = c.random_fr_array(bn)
        i0 = c.random_fr_array(1 << bn)
        want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
        for k in (0, 4, 8):
            gk.profile_reset(0)
            gk.set_option("test_corrupt_sum", k)
            assert np.array_equal(s.prove(qp), want), ("round", k)
            assert gk.profile_get()["layer_check_failures"] == 1, (k, gk.profile_get())
        print("INTEGRITY-OK")
    """, {"GKRHIP_GENERIC": "1"})


def test_without_the_check_the_same_slip_returns_a_wrong_transcript():
    """The control: layer_check = 0 and the same flipped bit -> rc 0 and a transcript that differs from the oracle's and that
    gkr.Verify rejects -- which is what verify_after_prove turns into an error on the one-shot path (hints.go:224-228)."""
    _run("""
        bn = 11
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
        qp = c.random_fr_array(bn)
        i0 = c.random_fr_array(1 << bn)
        want, outs, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
        gk.set_option("layer_check", 0)
        gk.profile_reset(0)
        gk.set_option("test_corrupt_sum", 2)
        got = s.prove(qp)
        assert not np.array_equal(got, want), "the hook did not corrupt anything"
        assert not s.verify(qp, got)
        assert c.gkr_verify_mimc(bn, got, i0, i0, outs, qp) != 0
        assert gk.profile_get()["layer_checks"] == 0
        assert np.array_equal(s.prove(qp), want)
        # the one-shot call, as the hint issues it: verify_after_prove catches what the (disabled) layer check would have
        gk.set_option("verify_after_prove", 1)
        flat, o2 = gk.gkr_prove_mimc(i0, i0.copy(), qp)
        assert np.array_equal(flat, want) and np.array_equal(o2, outs)
        gk.set_option("test_corrupt_sum", 2)
        try:
            gk.gkr_prove_mimc(i0, i0.copy(), qp)
            raise SystemExit("a wrong proof was returned")
        except gk.GkrHipError as e:
            assert "GKR proof was wrong" in str(e), e
        gk.set_option("layer_check", 1)
        flat, _ = gk.gkr_prove_mimc(i0, i0.copy(), qp)
        assert np.array_equal(flat, want)
        print("INTEGRITY-OK")
    """)


def test_a_slip_that_repeats_is_an_error_not_a_proof():
    """The flip fires twice: the safe-mode retry does not close either -> error return, no transcript."""
    _run("""
        bn = 10
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
        qp = c.random_fr_array(bn)
        i0 = c.random_fr_array(1 << bn)
        want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
        gk.set_option("test_corrupt_sum", 1)
        gk.set_option("test_corrupt_times", 2)
        try:
            s.prove(qp)
            raise SystemExit("a proof was returned")
        except gk.GkrHipError as e:
            assert "failed twice" in str(e), e
        for _ in range(2):
            assert np.array_equal(s.prove(qp), want)
        print("INTEGRITY-OK")
    """)


def test_twelve_lanes_one_slip():
    """Twelve proofs in flight, the solo paths forced on (the configuration that produced the unexplained wrong proofs of
    round 4), one flipped bit somewhere: every transcript is the oracle's, exactly one layer was run again."""
    _run("""
        bn, lanes = 13, 12
        qp = c.random_fr_array(bn)
        i0 = c.random_fr_array(1 << bn)
        want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
        ss = []
        for _ in range(lanes):
            s = gk.MimcSession(bn); s.synth_inputs(); s.assign(); ss.append(s)
        bad = []
        def work(k):
            for rep in range(4):
                if not np.array_equal(ss[k].prove(qp), want):
                    bad.append((k, rep))
        gk.profile_reset(0)
        gk.set_option("test_corrupt_sum", 3)
        ths = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not bad, bad
        p = gk.profile_get()
        assert p["layer_check_failures"] == 1 and p["layer_checks"] == lanes * 4 * 92, p
        print("INTEGRITY-OK")
    """, {"GKRHIP_PRELAUNCH": "2", "GKRHIP_PRE": "2", "GKRHIP_SPEC": "2"})


def test_sumcheck_prove_with_outside_claims_is_checked_too():
    """sumcheck.Prove on host tables (claims from outside only feed Fiat-Shamir): single point (fused rounds, every sum
    computed) and multi-instance (reference-shaped rounds); a flipped bit is caught from the next round's check on."""
    _run("""
        bn = 10
        n = 1 << bn
        rng = np.random.default_rng(7)
        X = [c.random_fr_array(n), c.from_ints([int(v) for v in rng.integers(0, 1 << 62, n)])]
        ark = c.from_u64(145646)
        for ninst in (1, 3):
            qs = np.stack([c.random_fr_array(bn) if j == 0 else c.from_ints([(j * 977 + i * i) for i in range(bn)]) for j in range(ninst)])
            claims = np.concatenate([c.evaluation(c.GATE_CIPHER, ark, qs[j:j + 1], c.fr(0), X) for j in range(ninst)])
            want = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims)
            for k in (None, 0, 4):
                gk.profile_reset(0)
                if k is not None:
                    gk.set_option("test_corrupt_sum", k)
                got = gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), (ninst, k)
                p = gk.profile_get()
                assert p["layer_checks"] == 1 and p["layer_check_failures"] == (0 if k is None else 1), (ninst, k, p)
            # claims that are NOT the sums are the caller's business (the reference only hashes them): no false alarm
            bogus = claims.copy(); bogus[0, 0] ^= np.uint64(1)
            gk.profile_reset(0)
            got = gk.sumcheck_prove(X, qs, bogus, gk.GATE_CIPHER, ark)
            for a, b in zip(got, c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, bogus)):
                assert np.array_equal(a, b), ("bogus", ninst)
            assert gk.profile_get()["layer_check_failures"] == 0
        print("INTEGRITY-OK")
    """)


def test_gmimc_circuit_linear_layers():
    """The GMiMC (t = 2) circuit: cipher, add-constant and identity layers (k_linear_round, two sums per round) with a slip."""
    _run("""
        import pyoracle as o
        bn = 11
        n = 1 << bn
        layers = gk.gmimc_t2_circuit()
        descs = c.circuit_descs(o.gmimc_t2_circuit())
        ins = [c.random_fr_array(n) if i % 2 == 0 else c.from_ints([(7 * j * j + i) % 1000003 for j in range(n)]) for i in range(4)]
        qp = c.random_fr_array(bn)
        s = gk.MimcSession(bn, layers=layers)
        for i, t in enumerate(ins):
            s.load_input(i, t)
        s.assign()
        want = c.gkr_prove_circuit(descs, bn, ins, qp)[0]
        assert np.array_equal(s.prove(qp), want)
        gk.profile_reset(0)
        assert np.array_equal(s.prove(qp), want)
        per = gk.profile_get()["layer_checks"]
        for k in (0, 2, 4):
            for layer in range(0, per, 7):          # walk the slip through the layer kinds
                gk.profile_reset(0)
                gk.set_option("test_corrupt_sum", k)
                gk.set_option("test_corrupt_skip", layer)
                assert np.array_equal(s.prove(qp), want), (k, layer)
                assert gk.profile_get()["layer_check_failures"] == 1, (k, layer, gk.profile_get())
        print("INTEGRITY-OK")
    """)


ARENA = """
    gk.set_option("arena_check", 1)
    bn, lanes = %d, %d
    qp = c.random_fr_array(bn)
    i0 = c.random_fr_array(1 << bn)
    want, wouts, _ = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
    ss = []
    for _ in range(lanes):
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign(); ss.append(s)
    bad = []
    def work(k):
        for rep in range(3):
            if not np.array_equal(ss[k].prove(qp), want):
                bad.append((k, rep))
        # the one-shot call of the hint (a session, a lane and a download lane per call) between the resident proofs
        flat, outs = gk.gkr_prove_mimc(i0, i0.copy(), qp)
        if not (np.array_equal(flat, want) and np.array_equal(outs, wouts)):
            bad.append((k, "oneshot"))
    ths = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for s in ss:
        s.close()
    assert not bad, bad
    n = gk.profile_counter("arena_busy_releases")
    assert n == 0, "%%d tables went back to the arena while their lane's stream still had work queued (sites on stderr)" %% n
    print("INTEGRITY-OK")
"""


def test_no_table_goes_back_to_the_arena_while_its_lane_is_busy():
    """The arena hands a released buffer to the next caller of its size class -- since round 5 the most recently released one
    first, often another lane with another stream.  With arena_check on, every release asks the releasing lane's streams
    whether they are idle: one proof alone (everything queued ahead), the solo paths forced on for eight lanes, the
    reference-shaped evaluator, and one-shot calls in between; transcripts against the oracle as everywhere."""
    _run(ARENA % (12, 1))
    _run(ARENA % (13, 8), {"GKRHIP_PRELAUNCH": "2", "GKRHIP_PRE": "2", "GKRHIP_SPEC": "2"})
    _run(ARENA % (13, 8))
    _run(ARENA % (11, 4), {"GKRHIP_GENERIC": "1"})
