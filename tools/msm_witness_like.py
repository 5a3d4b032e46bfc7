"""MSM on a witness-like scalar vector (most wires are 0 or 1, the rest full-width): the bucket of digit 1 in window 0 holds a
large share of all points.  Wall clock of gkrhip_msm_g1 (bases resident, scalars from host memory).  python tools/msm_witness_like.py [logn...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
G1 = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f,
               0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)
for logn in [int(a) for a in sys.argv[1:]] or [20, 22]:
    n = 1 << logn
    rng = np.random.default_rng(logn)
    k = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    b = gk.G1Bases(base=G1, scalars=k)
    full = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    full[:, 3] &= np.uint64((1 << 60) - 1)
    u = rng.random(n)
    wit = full.copy()
    wit[u < 0.4] = 0                                   # 40 % zeros
    wit[(u >= 0.4) & (u < 0.8)] = np.array([1, 0, 0, 0], dtype=np.uint64)      # 40 % ones
    for name, s in (("uniform", full), ("witness-like (40 % 0, 40 % 1)", wit)):
        b.multi_exp(s)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            b.multi_exp(s)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("2^%d %s: %.2f ms per call (median of 5), scalars' upload included" % (logn, name, 1e3 * ts[2]))
    b.close()
