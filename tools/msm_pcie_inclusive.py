"""The MSM as the cgo shim calls it: bases resident (uploaded once), scalars from host memory every call (gkrhip_msm_g1 /
gkrhip_msm_g2) -- pageable memory, and page-locked memory of gkrhip_host_alloc.  Wall clock per call, PCIe-inclusive.
python tools/msm_pcie_inclusive.py [logn...]"""
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
    s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    for name, cls, base in (("G1", gk.G1Bases, G1), ("G2", gk.G2Bases, gk.g2_generator())):
        t0 = time.perf_counter()
        b = cls(base=base, scalars=k)
        t_gen = time.perf_counter() - t0
        with gk.PinnedArray(n, 4) as pin:
            pin.a[:] = s
            for kind, sv in (("pageable", s), ("page-locked", pin.a)):
                b.multi_exp(sv)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    b.multi_exp(sv)
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                print("%s 2^%d, %s scalars: %.2f ms per call incl. the upload (median of 5; min %.2f) = %.0f M points/s; bases generated on the device in %.0f ms"
                      % (name, logn, kind, 1e3 * ts[2], 1e3 * ts[0], n / ts[2] / 1e6, 1e3 * t_gen))
        b.close()
