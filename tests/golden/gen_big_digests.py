#!/usr/bin/env python3
"""Generate tests/golden/gkr_mimc_big_digests.json: SHA-256 digests of the flat GKR proof (and of the
output table) produced by the C ORACLE (oracle/gkr_oracle.c, multi-threaded CPU restatement of the
reference) for the BASELINE sizes bN = 20, 22, 24 with RandomFrArray inputs (gkr/gkr_test.go:23-25).
Takes minutes of CPU time (about 3 minutes for bN = 24 on 16 cores); it was run once on the GPU box's
host CPU and its output committed.  The GPU parity tests compare the HIP prover's digests with these.
`gen_big_digests.py gmimc [sizes]` does the same for the GMiMC (t = 2) circuit of BASELINE config 5
(bN = 14, 20, 22 -> gkr_gmimc_big_digests.json; run in the build container, 8 cores)."""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import coracle as c  # noqa: E402


def gmimc(sizes, out_path):
    """Same for the build-defined GMiMC (t = 2) circuit (BASELINE config 5): every input layer =
    RandomFrArray(2^bN), which is what gkrhip_mimc_session_synth_inputs generates on the device."""
    import pyoracle as o
    descs = c.circuit_descs(o.gmimc_t2_circuit())
    res = []
    for bn in sizes:
        t0 = time.time()
        i0 = c.random_fr_array(1 << bn)
        ins = [i0, i0, i0, i0]
        qp = c.random_fr_array(bn)
        flat, outs, secs = c.gkr_prove_circuit(descs, bn, ins, qp)
        rc = c.gkr_verify_circuit(descs, bn, flat, ins, outs, qp)
        assert rc == 0, rc
        res.append({"bn": bn, "circuit": "gmimc_t2", "n_elements": int(flat.shape[0]),
                    "sha256_flat": hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest(),
                    "sha256_outputs": hashlib.sha256(outs.astype("<u8").tobytes()).hexdigest(),
                    "oracle_prove_seconds": secs, "oracle_threads": c.lib.oracle_num_threads(),
                    "verified_by_oracle_gkr_verify": True})
        print("gmimc", bn, "done in %.1f s (prove %.1f s)" % (time.time() - t0, secs), flush=True)
        json.dump(res, open(out_path, "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "gmimc":
        sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "14,20,22").split(",")]
        return gmimc(sizes, sys.argv[3] if len(sys.argv) > 3 else os.path.join(HERE, "gkr_gmimc_big_digests.json"))
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,22,24").split(",")]
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "gkr_mimc_big_digests.json")
    res = []
    for bn in sizes:
        t0 = time.time()
        i0 = c.random_fr_array(1 << bn)
        qp = c.random_fr_array(bn)
        flat, outs, secs = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)
        rc = c.gkr_verify_mimc(bn, flat, i0, i0, outs, qp)
        assert rc == 0, rc
        res.append({"bn": bn, "n_elements": int(flat.shape[0]),
                    "sha256_flat": hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest(),
                    "sha256_outputs": hashlib.sha256(outs.astype("<u8").tobytes()).hexdigest(),
                    "oracle_prove_seconds": secs, "oracle_threads": c.lib.oracle_num_threads(),
                    "verified_by_oracle_gkr_verify": True})
        print(bn, "done in %.1f s (prove %.1f s)" % (time.time() - t0, secs), flush=True)
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
