"""Device-resident fold micro-benchmark (BenchmarkFolding shape, poly/multilin_test.go:55-78): variants
interleaved in ONE process, several rounds, median and best reported (MI355X guide rule: perf deltas come
from interleaved rounds in one process).  Also what is run under rocprofv3 for the fold kernel's PMC counters."""
import importlib
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
variants = [("split", 65536), ("split", 32768), ("split", 8192), ("fused", 65536), ("fused", 8192)]
if quick:
    variants = variants[:1]
res = {}
for rnd in range(1 if quick else 5):
    for split, grid in variants:
        gk.set_option("fold_split", 1 if split == "split" else 0)
        gk.set_option("fold_grid", grid)
        for ntab in (1, 3):
            ms = gk.bench_fold(1 << bn, ntab=ntab, warmup=2, iters=10)
            res.setdefault((split, grid, ntab), []).append(ms)
for (split, grid, ntab), v in res.items():
    gbs = lambda ms: 96.0 * ntab * (1 << (bn - 1)) / (ms * 1e-3) / 1e9
    print("fold 2^%d x %d tables  %-5s grid<=%-6d median %.4f ms = %6.1f GB/s   best %.4f ms = %6.1f GB/s"
          % (bn, ntab, split, grid, statistics.median(v), gbs(statistics.median(v)), min(v), gbs(min(v))))
