"""computeH on device-resident vectors (gkrhip_bench_compute_h): python tools/computeh_bench.py [logn...] [--iters N]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
iters = 3
args = sys.argv[1:]
if "--iters" in args:
    iters = int(args[args.index("--iters") + 1])
    del args[args.index("--iters"):args.index("--iters") + 2]
for logn in [int(a) for a in args] or [20, 22, 24]:
    ms, p, by = gk.bench_compute_h(logn, 1, iters)
    print("2^%d: %.3f ms, %d passes, %.1f GB moved, %.0f GB/s" % (logn, ms, p, by / 1e9, by / ms / 1e6))
