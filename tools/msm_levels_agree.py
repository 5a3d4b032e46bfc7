"""The one- and the two-level sort give the same point at sizes no oracle reaches: python tools/msm_levels_agree.py [logn...]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
for logn in [int(a) for a in sys.argv[1:]] or [24, 26]:
    res = {}
    for lv in (1, 2):
        gk.set_option("msm_sort_levels", lv)
        r = gk.bench_msm_g1(logn, warmup=0, iters=1)
        res[lv] = r
    same = res[1]["result"].tolist() == res[2]["result"].tolist()
    print("2^%d: c = %d, one level %.2f ms, two levels %.2f ms, same point: %s" % (logn, res[1]["c"], res[1]["ms"], res[2]["ms"], same))
    assert same and any(res[1]["result"].tolist())
