"""Fixed-base MSM against the per-window path: python tools/r6_msm_fb.py [logn ...]"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
for lg in [int(a) for a in sys.argv[1:]] or [16, 20, 22, 24]:
    base = gk.bench_msm_g1(lg, warmup=1, iters=3)
    print("2^%d per-window c=%d: %.3f ms  %s" % (lg, base["c"], base["ms"], {k: round(v, 3) for k, v in base["phases_ms"].items()}), flush=True)
    for c in ([0] if lg < 20 else [20, 22]):
        r = gk.bench_msm_g1_fixed_base(lg, c=c, warmup=1, iters=3)
        assert r["result"].tolist() == base["result"].tolist(), "fixed-base result differs"
        print("2^%d fixed-base c=%d: %.3f ms  %s  tables %.0f ms" % (lg, r["c"], r["ms"], {k: round(v, 3) for k, v in r["phases_ms"].items()}, r["precompute_ms"]), flush=True)
