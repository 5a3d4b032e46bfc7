"""PCIe-inclusive rate of the one-shot entry point gkrhip_gkr_prove_mimc (what GkrProverHint.Call would bind,
prover/gadget/hints.go:220-222): host AoS inputs -> upload + limb-plane transposition + Circuit.Assign on the
device + gkr.Prove + download of the output table.  Reported in DESIGN.md; never bench.py's `value`."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_fr_array_np  # noqa: E402

gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << bn
rng = np.random.default_rng(1)
ins = []
for _ in range(2):   # any canonical residues serve as inputs for a timing
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    ins.append(a)
qp = random_fr_array_np(bn)
for rep in range(3):
    t0 = time.perf_counter()
    flat, outs = gk.gkr_prove_mimc(ins[0], ins[1], qp)
    dt = time.perf_counter() - t0
    print("bN=%d  one-shot gkr_prove_mimc from host buffers: %.3f s  -> %.2f M hashes/s (upload 2 x %d MiB, assign, prove, "
          "download %d MiB)" % (bn, dt, n / dt / 1e6, 32 * n >> 20, 32 * n >> 20), flush=True)
ok = gk.gkr_verify_mimc(flat, ins[0], ins[1], outs, qp)
print("gkr.Verify on the last proof:", ok)
# the same on regular-form buffers (gkrhip_gkr_prove_mimc_regular: the hint's big.Int words, conversions on the device)
for rep in range(2):
    t0 = time.perf_counter()
    flat_r, outs_r = gk.gkr_prove_mimc(ins[0], ins[1], qp, regular=True)
    dt = time.perf_counter() - t0
    print("bN=%d  one-shot gkr_prove_mimc_regular (regular-form buffers): %.3f s  -> %.2f M hashes/s" % (bn, dt, n / dt / 1e6), flush=True)
