"""Helper run in a subprocess by the GPU tests with different GKRHIP_* environment settings (they are
read once, at library initialisation): compares sumcheck.Prove / gkr.Prove on the GPU against the
oracle for a list of sizes and prints CASE-OK."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import coracle as c  # noqa: E402


def check_expected_paths(gk):
    """GKRHIP_CASE_EXPECT=prelaunched_rounds,lookahead_round0,...: the named serial-latency paths must have run (a switch
    that silently selects nothing would make the comparison above vacuous); GKRHIP_CASE_EXPECT_NOT: must not have run."""
    prof = gk.profile_get()
    for name in filter(None, os.environ.get("GKRHIP_CASE_EXPECT", "").split(",")):
        assert prof[name] > 0, (name, prof)
    for name in filter(None, os.environ.get("GKRHIP_CASE_EXPECT_NOT", "").split(",")):
        assert prof[name] == 0, (name, prof)


def main():
    sizes = [int(x) for x in sys.argv[1].split(",")]
    circuit = sys.argv[2] if len(sys.argv) > 2 else "mimc"
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    # knobs that are library OPTIONS, not environment switches (round 6 folded eight of them): "key=value,key=value"
    for kv in filter(None, os.environ.get("GKR_CASE_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        gk.set_option(k, int(v))
    gk.profile_reset(0)
    if circuit == "gmimc":      # the build-defined GMiMC (t = 2) circuit: cipher, add and identity layers
        import pyoracle as o
        layers = gk.gmimc_t2_circuit()
        descs = c.circuit_descs(o.gmimc_t2_circuit())
        for bn in sizes:
            n = 1 << bn
            ins = [c.random_fr_array(n) if i % 2 == 0 else c.from_ints([(7 * j * j + i) % 1000003 for j in range(n)]) for i in range(4)]
            qp = c.random_fr_array(bn)
            s = gk.MimcSession(bn, layers=layers)
            for i, t in enumerate(ins):
                s.load_input(i, t)
            s.assign()
            flat = s.prove(qp)
            oflat, _oouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
            assert np.array_equal(flat, oflat), ("gmimc", bn)
            s.close()
        check_expected_paths(gk)
        print("CASE-OK", sizes)
        return
    for bn in sizes:
        n = 1 << bn
        rng = np.random.default_rng(bn)
        X = [c.random_fr_array(n), c.from_ints([int(v) for v in rng.integers(0, 1 << 62, n)])]
        ark = c.from_u64(145646)
        qs = c.random_fr_array(bn).reshape(1, bn, 4)
        claims = c.evaluation(c.GATE_CIPHER, ark, qs, c.fr(0), X)
        got = gk.sumcheck_prove(X, qs, claims, gk.GATE_CIPHER, ark)
        want = c.sumcheck_prove(c.GATE_CIPHER, ark, X, qs, claims)
        for a, b in zip(got, want):
            assert np.array_equal(a, b), ("sumcheck", bn)
        i0, qp = c.random_fr_array(n), c.random_fr_array(bn)
        flat, outs = gk.gkr_prove_mimc(i0, X[1], qp)
        oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, X[1], qp)
        assert np.array_equal(flat, oflat) and np.array_equal(outs, oouts), ("gkr", bn)
        nsess = int(os.environ.get("GKRHIP_CASE_SESSIONS", "0"))
        if nsess:       # several resident sessions (a lane and stream each) proving at once, every transcript against the oracle's
            import threading
            ss = []
            for _ in range(nsess):
                sn = gk.MimcSession(bn)
                sn.load_inputs(i0, X[1])
                sn.assign()
                ss.append(sn)
            res = [None] * nsess

            def work(k):
                for _ in range(3):
                    res[k] = ss[k].prove(qp)
            ths = [threading.Thread(target=work, args=(k,)) for k in range(nsess)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for k in range(nsess):
                assert np.array_equal(res[k], oflat), ("session", k, bn)
                ss[k].close()
    check_expected_paths(gk)
    print("CASE-OK", sizes)


if __name__ == "__main__":
    main()
