"""Proof groups (gkrhip_mimc_session_prove_group): parity with the single proofs, then throughput of T host threads x groups of k
against the same number of proofs in flight on lanes of their own.
python tools/r6_group_probe.py [gmimc] [bn] [in_flight] [k,k,...] [proofs_per_session] [g_max,g_max,...]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
GMIMC = "gmimc" in sys.argv      # the GMiMC t = 2 circuit (BASELINE config 5) instead of examples.MimcCircuit
sys.argv = [a for a in sys.argv if a != "gmimc"]
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nfl = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ks = [x for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]      # "3": explicit groups of 3; "1": single calls, never grouped; "c3": single calls, the library forms groups of 3 (option group_size)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
gmaxes = [None if x in ("-", "None") else int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [None]
for kv in os.environ.get("GKRHIP_BENCH_OPTIONS", "").split(","):
    if "=" in kv:
        gk.set_option(kv.split("=")[0], int(kv.split("=")[1]))
rng = np.random.default_rng(5)


def rnd_q():
    a = rng.integers(0, 1 << 63, size=(bn, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


sessions = []
for i in range(nfl):
    s = gk.MimcSession(bn, layers=gk.gmimc_t2_circuit() if GMIMC else None)
    s.synth_inputs(stride=1, offset=i * 7919)      # every session its own inputs
    s.assign()
    sessions.append(s)
qs = [rnd_q() for _ in range(nfl)]

# parity: groups of every size up to 8 against the single proofs
gk.set_option("group_size", 0)
single = [s.prove(q) for s, q in zip(sessions[:8], qs[:8])]
for k in (1, 2, 3, 5, 8):
    if k > nfl:
        continue
    got = gk.MimcSession.prove_group(sessions[:k], qs[:k])
    same = [bool(np.array_equal(a, b)) for a, b in zip(got, single)]
    ok = [sessions[i].verify(qs[i], got[i]) for i in range(k)]
    print("group of %d: identical to the single proofs %s, gkr.Verify %s" % (k, same, ok), flush=True)
    assert all(same) and all(ok)


def run(k):
    coalesce = 0
    if str(k).startswith("c"):
        coalesce, k = int(k[1:]), 1
    k = int(k)
    gk.set_option("group_size", coalesce)
    gk.profile_reset(1 << 40)
    chunks = [list(range(i, min(i + k, nfl))) for i in range(0, nfl, k)]
    errs = []

    def work(idx):
        try:
            for _ in range(reps):
                if k == 1:
                    sessions[idx[0]].prove(qs[idx[0]])
                else:
                    gk.MimcSession.prove_group([sessions[i] for i in idx], [qs[i] for i in idx])
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    best = None
    for trial in range(3):
        ths = [threading.Thread(target=work, args=(c,)) for c in chunks]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        if errs:
            raise errs[0]
        if trial:
            best = dt if best is None else min(best, dt)
    per = best / (reps * nfl)
    what = ("single calls, groups of %d formed by the library (%d of %d proofs)" % (coalesce, gk.profile_counter("coalesced_proofs"), 3 * reps * nfl)) if coalesce else "groups of %d" % k
    print("bN=%d, %d proofs in flight, %s (%d host threads)%s: %.2f ms per proof, %.2f M hashes/s" %
          (bn, nfl, what, len(chunks), "" if gm is None else ", g_max %d" % gm, 1e3 * per, (1 << bn) / per / 1e6), flush=True)


sweeps = sys.argv[6].split(";") if len(sys.argv) > 6 else [""]      # e.g. "slim=0;slim=1,slim_lg=14": option sets, one pass each
for sw in sweeps:
    for kv in filter(None, sw.split(",")):
        gk.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if sw:
        print("--- options:", sw, flush=True)
    for gm in gmaxes:
        if gm is not None:
            gk.set_option("g_max", gm)
        for k in ks:
            run(k)
