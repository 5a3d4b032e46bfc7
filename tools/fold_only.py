"""Runs the device-resident fold micro-benchmark only (BenchmarkFolding shape, poly/multilin_test.go:55-78):
used under rocprofv3 to collect the fold kernel's duration and PMC counters."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for ntab in (1, 3):
    ms = gk.bench_fold(1 << bn, ntab=ntab, warmup=2, iters=10)
    gbs = 96.0 * ntab * (1 << (bn - 1)) / (ms * 1e-3) / 1e9
    print("fold 2^%d x %d tables: %.4f ms per launch, %.1f GB/s algorithmic (96 B per output element per table)" % (bn, ntab, ms, gbs))
