"""Which window size is fastest at which size: python tools/msm_window_sweep.py [g2]  (device-resident synthetic data)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
g2 = "g2" in sys.argv[1:]
f = gk.bench_msm_g2 if g2 else gk.bench_msm_g1
for logn in (8, 10, 12, 14, 16, 17, 18, 19, 20, 21, 22, 23):
    row = {}
    for c in range(min(14, max(4, logn - 6)), 17):
        row[c] = f(logn, c=c, warmup=1, iters=3)["ms"]
    best = min(row, key=row.get)
    auto = f(logn, c=0, warmup=0, iters=1)["c"]
    print("2^%d: best c = %d (%.3f ms); automatic choice c = %d (%.3f ms); %s" % (logn, best, row[best], auto, row.get(auto, float("nan")),
          " ".join("%d:%.2f" % (c, row[c]) for c in sorted(row))))
