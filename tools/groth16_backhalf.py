"""The device side of ComputeGroth16Proof (prover/gadget/prove.go:100-306) at one size: computeH, then the MSMs over pk.G1.A,
pk.G1.B, pk.G1.Z (with h), pk.privKNotGkr and pk.G2.B -- one after the other and from five host threads at once (every call
leases a lane of its own).  Bases resident, scalars from host memory; synthetic data ([k_i]G bases, random scalars).
python tools/groth16_backhalf.py [logn] [--fixed-base]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
gk.reserve_lanes(5)      # the five calls at once lease a lane each
G1 = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f,
               0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)
FIXED_BASE = "--fixed-base" in sys.argv      # round 6: every key vector on fixed-base tables (gkrhip_msm_g1_precompute / _g2_precompute)
sys.argv = [a for a in sys.argv if a != "--fixed-base"]
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << logn
rng = np.random.default_rng(1)


def rnd():
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


k = rnd()
bases = {name: gk.G1Bases(base=G1, scalars=k) for name in ("A", "B1", "Z", "K")}
b2 = gk.G2Bases(base=gk.g2_generator(), scalars=k)
if FIXED_BASE:
    t0 = time.perf_counter()
    for h in list(bases.values()) + [b2]:
        h.precompute(0)
    print("fixed-base tables of the five key vectors: %.0f ms (once per key)" % (1e3 * (time.perf_counter() - t0)))
wires, a, b, c = rnd(), rnd(), rnd(), rnd()
jobs = {
    "computeH + krs2 (pk.G1.Z)": lambda: bases["Z"].compute_h_multi_exp(a, b, c),
    "ar (pk.G1.A)": lambda: bases["A"].multi_exp(wires),
    "bs1 (pk.G1.B)": lambda: bases["B1"].multi_exp(wires),
    "krs (pk.privKNotGkr)": lambda: bases["K"].multi_exp(wires),
    "Bs (pk.G2.B)": lambda: b2.multi_exp(wires),
}
for f in jobs.values():
    f()                                         # warm: work buffers, NTT domain
t_all = time.perf_counter()
for name, f in jobs.items():
    t0 = time.perf_counter()
    f()
    print("%-28s %.2f ms" % (name, 1e3 * (time.perf_counter() - t0)))
serial = time.perf_counter() - t_all


def at_once(fs):
    """wall time of the calls from one host thread each; the first round creates the lanes the calls lease (untimed)"""
    best = None
    for _ in range(3):
        ths = [threading.Thread(target=f) for f in fs]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        if _:                       # rounds 1 and 2 count
            best = dt if best is None else min(best, dt)
    return best


par = at_once(list(jobs.values()))
print("2^%d: one after the other %.1f ms, five host threads at once %.1f ms (uploads of the scalars and of a, b, c included)" % (logn, 1e3 * serial, 1e3 * par))
# bs1 and Bs share wireValuesB (prove.go:189,277): one upload, one sort
pair = {k: v for k, v in jobs.items() if not k.startswith(("bs1", "Bs"))}
pair["bs1 + Bs (pk.G1.B, pk.G2.B), one sort"] = lambda: gk.multi_exp_g1_g2(bases["B1"], b2, wires)
pair["bs1 + Bs (pk.G1.B, pk.G2.B), one sort"]()
t_all = time.perf_counter()
for name, f in pair.items():
    t0 = time.perf_counter()
    f()
    print("%-40s %.2f ms" % (name, 1e3 * (time.perf_counter() - t0)))
print("2^%d with the paired call: %.1f ms" % (logn, 1e3 * (time.perf_counter() - t_all)))
# the same with every per-proof vector in page-locked memory (gkrhip_host_alloc)
pins = [gk.PinnedArray(n, 4) for _ in range(4)]
for pa, src in zip(pins, (wires, a, b, c)):
    pa.a[:] = src
pw, pa_, pb_, pc_ = (x.a for x in pins)
pinned = {
    "computeH + krs2 (pk.G1.Z)": lambda: bases["Z"].compute_h_multi_exp(pa_, pb_, pc_),
    "ar (pk.G1.A)": lambda: bases["A"].multi_exp(pw),
    "krs (pk.privKNotGkr)": lambda: bases["K"].multi_exp(pw),
    "bs1 + Bs (pk.G1.B, pk.G2.B), one sort": lambda: gk.multi_exp_g1_g2(bases["B1"], b2, pw),
}
for f in pinned.values():
    f()
t_all = time.perf_counter()
for name, f in pinned.items():
    t0 = time.perf_counter()
    f()
    print("%-40s %.2f ms (page-locked inputs)" % (name, 1e3 * (time.perf_counter() - t0)))
serial = time.perf_counter() - t_all
# ar too on the shared sort (pk.G1.A expanded with points at infinity so that it lines up with the unfiltered wireValues)
shared3 = lambda: gk.multi_exp_shared([bases["A"], bases["B1"]], [b2], pw)      # noqa: E731
shared3()
t0 = time.perf_counter()
shared3()
print("%-40s %.2f ms (page-locked inputs)" % ("ar + bs1 + Bs, one sort", 1e3 * (time.perf_counter() - t0)))
par = at_once(list(pinned.values()))
print("2^%d, paired call, page-locked inputs: one after the other %.1f ms, four host threads at once %.1f ms" % (logn, 1e3 * serial, 1e3 * par))
par3 = at_once([pinned["computeH + krs2 (pk.G1.Z)"], pinned["krs (pk.privKNotGkr)"], shared3])
print("2^%d, ar + bs1 + Bs on one sort, page-locked inputs, three host threads at once: %.1f ms" % (logn, 1e3 * par3))
