"""GPU tests (-m gpu) of proof groups (gkrhip_mimc_session_prove_group, host_group.hip.h): n sessions proven in lock-step by one host
thread, their round kernels launched together.  Every proof of a group must be, bit for bit, the transcript of gkr.Prove
(gkr/prover.go:21-91) for its own inputs and point -- checked against the oracle (oracle/coracle.py) at the sizes it finishes in
seconds, against the single call and the native gkr.Verify above them -- whatever the other proofs of the group do: other inputs,
other sizes (their launches then do not merge), a slip in one of them that sends it through the safe-mode rerun, a second group
running beside the first."""
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
    rng = np.random.default_rng(11)
    def rnd(n):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    def session(bn, layers=None):
        s = gk.MimcSession(bn, layers=layers)
        ins = [rnd(1 << bn) for _ in range(s.num_inputs)]
        for i, t in enumerate(ins):
            s.load_input(i, t)
        s.assign()
        return s, ins
""") % (ROOT, os.path.join(ROOT, "oracle"))


def _run(body, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "GROUPS-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_group_proofs_are_the_oracles_transcripts():
    """Groups of 1, 2, 3, 5 and 8 sessions with inputs and points of their own, bN = 7 and 11: the oracle's transcript for each."""
    _run("""
        for bn in (7, 11):
            made = [session(bn) for _ in range(8)]
            qs = [rnd(bn) for _ in range(8)]
            want = [c.gkr_prove_mimc(bn, ins[0], ins[1], q)[0] for (s, ins), q in zip(made, qs)]
            for n in (1, 2, 3, 5, 8):
                gk.profile_reset(1 << 40)
                got = gk.MimcSession.prove_group([m[0] for m in made[:n]], qs[:n])
                for i in range(n):
                    assert np.array_equal(got[i], want[i]), (bn, n, i)
                w, m = gk.profile_counter("group_launches_wanted"), gk.profile_counter("group_launches_made")
                # same shape: every launch served the whole group (the pyramids' launch carries its coordinates and holds three proofs:
                # a larger group takes two or three of those)
                assert w > 0 and (w == n * m if n <= 3 else 3 * m < w < n * m), (bn, n, w, m)
                assert gk.profile_get()["layer_check_failures"] == 0
            for s, _ in made:
                s.close()
        print("GROUPS-OK")
    """)


def test_group_at_the_sizes_of_the_wide_kernels():
    """bN = 16 and 18 (several pairs per lane: the deferred-reduction kernels, round 0 ahead of its point, the host tail): the group's
    proofs are the single calls' and gkr.Verify accepts them; again with the group's thread count forced."""
    body = """
        for bn, n in ((16, 4), (18, 3)):
            made = [session(bn) for _ in range(n)]
            qs = [rnd(bn) for _ in range(n)]
            single = [m[0].prove(q) for m, q in zip(made, qs)]
            for rep in range(2):
                got = gk.MimcSession.prove_group([m[0] for m in made], qs)
                for i in range(n):
                    assert np.array_equal(got[i], single[i]), (bn, rep, i)
                    assert made[i][0].verify(qs[i], got[i]), (bn, rep, i)
            for s, _ in made:
                s.close()
        print("GROUPS-OK")
    """
    _run(body)
    _run("gk.set_option('g_max', 10)\n" + textwrap.dedent(body))


def test_group_of_different_shapes_and_circuits():
    """Sessions of different sizes in one group (their launches cannot merge: each gets its own), and a group over the GMiMC circuit
    (linear and cipher layers; its batch holds six proofs: a group of eight takes two launches)."""
    _run("""
        made = [session(bn) for bn in (9, 12, 9, 6)]
        qs = [rnd(bn) for bn in (9, 12, 9, 6)]
        want = [c.gkr_prove_mimc(bn, m[1][0], m[1][1], q)[0] for bn, m, q in zip((9, 12, 9, 6), made, qs)]
        got = gk.MimcSession.prove_group([m[0] for m in made], qs)
        for i in range(4):
            assert np.array_equal(got[i], want[i]), i
        for s, _ in made:
            s.close()
        layers = gk.gmimc_t2_circuit()
        bn = 10
        made = [session(bn, layers) for _ in range(8)]
        qs = [rnd(bn) for _ in range(8)]
        single = [m[0].prove(q) for m, q in zip(made, qs)]
        for n in (2, 6, 8):
            got = gk.MimcSession.prove_group([m[0] for m in made[:n]], qs[:n])
            for i in range(n):
                assert np.array_equal(got[i], single[i]), (n, i)
                assert made[i][0].verify(qs[i], got[i]), (n, i)
        print("GROUPS-OK")
    """)


def test_a_slip_in_one_proof_of_a_group_is_caught_and_the_others_are_untouched():
    """test_corrupt_sum flips a bit in ONE device sum of the process: that proof's layer does not close and is run again in safe mode,
    on launches of its own while the others wait at theirs; every proof still comes out as the oracle's."""
    _run("""
        bn = 11
        made = [session(bn) for _ in range(4)]
        qs = [rnd(bn) for _ in range(4)]
        want = [c.gkr_prove_mimc(bn, m[1][0], m[1][1], q)[0] for m, q in zip(made, qs)]
        for k, skip in ((0, 0), (2, 5), (4, 200), (1, 363)):
            gk.profile_reset(1 << 40)
            gk.set_option("test_corrupt_sum", k)
            gk.set_option("test_corrupt_skip", skip)
            got = gk.MimcSession.prove_group([m[0] for m in made], qs)
            for i in range(4):
                assert np.array_equal(got[i], want[i]), (k, skip, i)
            assert gk.profile_get()["layer_check_failures"] == 1, (k, skip, gk.profile_get())
        print("GROUPS-OK")
    """)


def test_two_groups_side_by_side_and_a_lane_beside_them():
    """Two host threads with a group each and a third proving single proofs, several rounds: everything identical to the single calls."""
    _run("""
        bn = 14
        made = [session(bn) for _ in range(7)]
        qs = [rnd(bn) for _ in range(7)]
        single = [m[0].prove(q) for m, q in zip(made, qs)]
        bad = []
        def grp(idx):
            for _ in range(4):
                got = gk.MimcSession.prove_group([made[i][0] for i in idx], [qs[i] for i in idx])
                for i, p in zip(idx, got):
                    if not np.array_equal(p, single[i]):
                        bad.append(i)
        def lane(i):
            for _ in range(6):
                if not np.array_equal(made[i][0].prove(qs[i]), single[i]):
                    bad.append(i)
        ths = [threading.Thread(target=grp, args=([0, 1, 2],)), threading.Thread(target=grp, args=([3, 4, 5],)), threading.Thread(target=lane, args=(6,))]
        for t in ths: t.start()
        for t in ths: t.join()
        assert not bad, bad
        print("GROUPS-OK")
    """)


def test_group_arguments_are_checked():
    _run("""
        s, _ = session(6)
        t, _ = session(6)
        q = rnd(6)
        for ss, what in (([s, s], "twice"), ([s] * 0 + [s, t] * 5, "twice")):
            try:
                gk.MimcSession.prove_group(ss, [q] * len(ss))
                raise SystemExit("accepted: " + what)
            except gk.GkrHipError as e:
                assert what in str(e) or "proofs" in str(e), str(e)
        many = [session(5)[0] for _ in range(9)]
        try:
            gk.MimcSession.prove_group(many, [rnd(5)] * 9)
            raise SystemExit("accepted nine")
        except gk.GkrHipError as e:
            assert "1..8" in str(e), str(e)
        got = gk.MimcSession.prove_group([s, t], [q, q])      # and the sessions still work
        assert np.array_equal(got[0], s.prove(q)) and np.array_equal(got[1], t.prove(q))
        print("GROUPS-OK")
    """)


def test_single_calls_that_meet_are_proven_in_groups():
    """The reference's call shape: a host thread per statement, each calling gkr.Prove.  From 24 callers with statements of 2^18..2^21
    entries on, the calls that arrive together are proven as a group by the first of them (option group_size, default 3): same
    transcripts, counter coalesced_proofs > 0; with the option at 0, with fewer callers or smaller statements nothing is grouped; a
    slip in one of the grouped proofs is caught as ever."""
    _run("""
        bn, nl = 18, 24
        made = [session(bn) for _ in range(nl)]
        qs = [rnd(bn) for _ in range(nl)]
        gk.set_option("group_size", 0)
        single = [m[0].prove(q) for m, q in zip(made, qs)]
        for i in (0, 13):
            assert made[i][0].verify(qs[i], single[i])
        def storm(reps, who=range(nl)):
            bad = []
            def work(i):
                for _ in range(reps):
                    if not np.array_equal(made[i][0].prove(qs[i]), single[i]):
                        bad.append(i)
            ths = [threading.Thread(target=work, args=(i,)) for i in who]
            for t in ths: t.start()
            for t in ths: t.join()
            return bad
        gk.profile_reset(1 << 40)
        assert not storm(2)
        assert gk.profile_counter("coalesced_proofs") == 0
        gk.set_option("group_size", 3)
        gk.profile_reset(1 << 40)
        assert not storm(4)
        n = gk.profile_counter("coalesced_proofs")
        assert 0 < n <= 4 * nl, n
        assert gk.profile_get()["layer_check_failures"] == 0
        gk.profile_reset(1 << 40)
        gk.set_option("test_corrupt_sum", 2)
        gk.set_option("test_corrupt_skip", 300)
        assert not storm(3)
        assert gk.profile_get()["layer_check_failures"] == 1, gk.profile_get()
        # twelve callers are left alone (a host-bound job: every caller needs its own core), one caller all the more
        gk.profile_reset(1 << 40)
        assert not storm(2, range(12))
        assert np.array_equal(made[3][0].prove(qs[3]), single[3])
        assert gk.profile_counter("coalesced_proofs") == 0
        print("GROUPS-OK")
    """)


def test_small_statements_from_many_callers_are_not_grouped():
    _run("""
        bn, nl = 12, 26
        made = [session(bn) for _ in range(nl)]
        qs = [rnd(bn) for _ in range(nl)]
        want = [c.gkr_prove_mimc(bn, m[1][0], m[1][1], q)[0] for m, q in zip(made[:3], qs[:3])]
        bad = []
        def work(i):
            for _ in range(6):
                p = made[i][0].prove(qs[i])
                if i < 3 and not np.array_equal(p, want[i]):
                    bad.append(i)
        gk.profile_reset(1 << 40)
        ths = [threading.Thread(target=work, args=(i,)) for i in range(nl)]
        for t in ths: t.start()
        for t in ths: t.join()
        assert not bad and gk.profile_counter("coalesced_proofs") == 0
        print("GROUPS-OK")
    """)
