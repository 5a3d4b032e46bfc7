#!/usr/bin/env python3
"""Many lanes proving ONE size at once (every lane's tables have the same capacities, so buffers migrate between lanes through
the arena and every lane runs the same kernels at the same time): each proof must equal the first one byte for byte.
    python tools/stress_one_size.py [bn] [lanes] [proofs per lane]
This is the load that exposed the look-ahead kernel's lowest-priority stream (round 4): with GKRHIP_PRELAUNCH=2 GKRHIP_PRE=2
(and pre-launch for every round size) and twelve lanes at bN = 18, 4 % of the proofs were wrong."""
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import random_fr_array_np  # noqa: E402


def main():
    bn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    ss = []
    for _ in range(lanes):
        s = gk.MimcSession(bn)
        s.synth_inputs()
        s.assign()
        ss.append(s)
    q = random_fr_array_np(bn)
    good = ss[0].prove(q)
    assert ss[0].verify(q, good)
    bad = []

    def work(i):
        for it in range(per):
            f = ss[i].prove(q)
            if not np.array_equal(f, good):
                bad.append((i, it, int(np.nonzero((f != good).any(axis=1))[0][-1])))

    th = [threading.Thread(target=work, args=(i,)) for i in range(lanes)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    p = gk.profile_get()
    print("bN = %d, %d lanes x %d proofs, mismatches (lane, proof, last differing row): %s; sumchecks checked %d, not closing %d, "
          "layers retried after a missed challenge %d, round 0 ahead %d, proven in groups formed from single calls %d"
          % (bn, lanes, per, bad, p["layer_checks"], p["layer_check_failures"], p["chal_retries"], p["ahead_round0"], gk.profile_counter("coalesced_proofs")))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
