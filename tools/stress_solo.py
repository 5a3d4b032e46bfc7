#!/usr/bin/env python3
"""Soak of the paths a proof takes when it is ALONE on the GPU (pre-launched rounds, look-ahead, cooperative and speculative
kernels -- the defaults): proofs of several sizes one at a time, every proof of a size byte-identical to the first one (which
the native verifier accepted).   python tools/stress_solo.py [seconds]"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import random_fr_array_np  # noqa: E402

def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    sizes = [7, 9, 10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22]
    sess, want, qps = {}, {}, {}
    for bn in sizes:
        s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
        sess[bn] = s
        qps[bn] = random_fr_array_np(bn)
        flat = s.prove(qps[bn])
        assert s.verify(qps[bn], flat), bn
        want[bn] = hashlib.sha256(flat.tobytes()).hexdigest()
    # the GMiMC circuit as well (keys 100 + bn): cipher layers and linear layers (their speculative rounds have two candidates)
    layers = gk.gmimc_t2_circuit()
    for bn in (8, 11, 14, 17):
        s = gk.MimcSession(bn, layers=layers)
        for i in range(4):
            s.load_input(i, random_fr_array_np(1 << bn) if i % 2 == 0 else random_fr_array_np(1 << bn)[::-1].copy())
        s.assign()
        key = 100 + bn
        sess[key], qps[key] = s, random_fr_array_np(bn)
        flat = s.prove(qps[key])
        assert s.verify(qps[key], flat), key
        want[key] = hashlib.sha256(flat.tobytes()).hexdigest()
    sizes = sizes + [100 + bn for bn in (8, 11, 14, 17)]
    gk.profile_reset(0)
    stop, n, bad = time.time() + budget, 0, []
    while time.time() < stop and not bad:
        for bn in sizes:
            if hashlib.sha256(sess[bn].prove(qps[bn]).tobytes()).hexdigest() != want[bn]:
                bad.append((bn, n))
            n += 1
    p = gk.profile_get()
    print("solo soak: %d proofs of sizes %s one at a time, mismatches: %s; speculative rounds %d, cooperative %d, pre-launched %d, look-ahead %d, "
          "round 0 ahead %d; sumchecks checked %d, not closing %d, layers retried after a missed challenge %d"
          % (n, sizes, bad, p["spec_rounds"], p["coop_rounds"], p["prelaunched_rounds"], p["lookahead_round0"], p["ahead_round0"],
             p["layer_checks"], p["layer_check_failures"], p["chal_retries"]))
    sys.exit(1 if bad else 0)

main()
