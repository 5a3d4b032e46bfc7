import importlib, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, "/root/repo/oracle")
gk = importlib.import_module("gkr-mimc_amd"); gk.init(0)
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << bn
rng = np.random.default_rng(1)
in0 = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); in0[:, 3] &= np.uint64((1 << 60) - 1)
in1 = in0[::-1].copy()
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import random_fr_array_np
qp = random_fr_array_np(bn)
for rep in range(3):
    t = [time.perf_counter()]
    s = gk.MimcSession(bn); t.append(time.perf_counter())
    s.load_input(0, in0); s.load_input(1, in1); gk.synchronize(); t.append(time.perf_counter())
    s.assign(); gk.synchronize(); t.append(time.perf_counter())
    flat = s.prove(qp); t.append(time.perf_counter())
    outs = s.outputs(); t.append(time.perf_counter())
    s.close(); t.append(time.perf_counter())
    names = ["create", "load", "assign", "prove", "outputs", "destroy"]
    print(rep, " ".join("%s=%.1fms" % (nm, 1e3 * (b - a)) for nm, a, b in zip(names, t, t[1:])), "total=%.1fms" % (1e3 * (t[-1] - t[0])))
