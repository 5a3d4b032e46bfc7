#!/usr/bin/env python3
"""Soak with GMiMC-circuit sessions on several lanes at once (cipher and linear layers, every size twice); each proof of a size must be
byte-identical to the first one.   python tools/stress_gmimc.py [seconds]"""
import hashlib, importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import random_fr_array_np  # noqa: E402

def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    layers = gk.gmimc_t2_circuit()
    sizes = [8, 10, 12, 14, 16, 18]
    sessions, want, qps, ins = [], {}, {}, {}
    for bn in sizes:
        ins[bn] = [random_fr_array_np(1 << bn) if i % 2 == 0 else random_fr_array_np(1 << bn)[::-1].copy() for i in range(4)]
        qps[bn] = random_fr_array_np(bn)
        for _ in range(2):
            s = gk.MimcSession(bn, layers=layers)
            for i in range(4):
                s.load_input(i, ins[bn][i])
            s.assign()
            sessions.append((bn, s))
        flat = sessions[-1][1].prove(qps[bn])
        assert sessions[-1][1].verify(qps[bn], flat), bn
        want[bn] = hashlib.sha256(flat.tobytes()).hexdigest()
    stop = time.time() + budget
    counts, bad = [0] * len(sessions), []

    def work(k):
        bn, s = sessions[k]
        while time.time() < stop and not bad:
            if hashlib.sha256(s.prove(qps[bn]).tobytes()).hexdigest() != want[bn]:
                bad.append((bn, k, counts[k]))
            counts[k] += 1

    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(sessions))]
    [t.start() for t in ths]; [t.join() for t in ths]
    print("gmimc soak: proofs per lane:", counts, "total", sum(counts), "mismatches:", bad, "spec rounds", gk.profile_get()["spec_rounds"], "layers retried after a missed challenge:", gk.profile_get()["chal_retries"], "sumchecks checked / not closing:", gk.profile_get()["layer_checks"], "/", gk.profile_get()["layer_check_failures"], "round 0 ahead:", gk.profile_get()["ahead_round0"], "proven in groups formed from single calls:", gk.profile_counter("coalesced_proofs"))
    sys.exit(1 if bad else 0)

main()
