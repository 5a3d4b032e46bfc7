#!/usr/bin/env python3
"""Soak test of the lanes / hand-off / atomic accumulation machinery: many proofs of several sizes from several
host threads at once; every proof of a size must be byte-identical to the first one (the prover is deterministic)
and the native verifier must accept it.   python tools/stress.py [seconds]"""
import hashlib
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import random_fr_array_np  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    sizes = [9, 12, 14, 16, 17, 18, 20]
    sessions, want, qps = [], {}, {}
    for bn in sizes:
        for _ in range(2):                       # two lanes per size
            s = gk.MimcSession(bn)
            s.synth_inputs()
            s.assign()
            sessions.append((bn, s))
        qps[bn] = random_fr_array_np(bn)
        flat = sessions[-1][1].prove(qps[bn])
        assert sessions[-1][1].verify(qps[bn], flat), bn
        want[bn] = hashlib.sha256(flat.tobytes()).hexdigest()
    stop = time.time() + budget
    counts, bad = [0] * len(sessions), []

    def work(k):
        bn, s = sessions[k]
        while time.time() < stop and not bad:
            flat = s.prove(qps[bn])
            if hashlib.sha256(flat.tobytes()).hexdigest() != want[bn]:
                bad.append((bn, k, counts[k]))
            counts[k] += 1

    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(sessions))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print("proofs per lane:", counts, "total", sum(counts), "mismatches:", bad, "layers retried after a missed challenge:", gk.profile_get()["chal_retries"], "sumchecks checked / not closing:", gk.profile_get()["layer_checks"], "/", gk.profile_get()["layer_check_failures"], "round 0 ahead:", gk.profile_get()["ahead_round0"], "proven in groups formed from single calls:", gk.profile_counter("coalesced_proofs"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
