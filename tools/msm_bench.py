"""MSM micro-benchmark on device-resident synthetic data (gkrhip_bench_msm_g1 / _g2): python tools/msm_bench.py [g2] [logn...] [c=N] [levels=1|2]"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
g2 = "g2" in sys.argv[1:]
args = [a for a in sys.argv[1:] if not a.startswith("c=") and not a.startswith("levels=") and a != "g2"]
levels = [int(a[7:]) for a in sys.argv[1:] if a.startswith("levels=")] or [0]
cs = [int(a[2:]) for a in sys.argv[1:] if a.startswith("c=")] or [0]
for logn in [int(a) for a in args] or [16, 18, 20, 22]:
    for cw, lv in [(cw, lv) for cw in cs for lv in levels]:
        gk.set_option("msm_sort_levels", lv)
        r = (gk.bench_msm_g2 if g2 else gk.bench_msm_g1)(logn, c=cw, warmup=1, iters=3)
        r["group"] = "G2" if g2 else "G1"
        r["sort_levels"] = lv
        r.pop("result")
        r["logn"] = logn
        r["points_per_s"] = (1 << logn) / (r["ms"] * 1e-3)
        print(json.dumps(r))
