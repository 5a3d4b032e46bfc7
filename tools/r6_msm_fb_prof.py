"""Only the fixed-base MSM at one size with the library's own sort, for rocprofv3 kernel statistics:
   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/r6_msm_fb_prof.py 24 22"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
lg, c = int(sys.argv[1]), int(sys.argv[2])
r = gk.bench_msm_g1_fixed_base(lg, c=c, warmup=1, iters=5)
print("2^%d c=%d: %.3f ms %s" % (lg, r["c"], r["ms"], {k: round(v, 3) for k, v in r["phases_ms"].items()}))
