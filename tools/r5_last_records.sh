#!/bin/bash
# Last records of the round-5 tree: smoke(), the GPU suite, and the default bench line N times (every run must print its line at
# the first attempt: the abort of profiles/r05_anomalies.md (c) showed as a missing line about once in twelve).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r5last
mkdir -p $OUT
cd $ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -1 $OUT/gpu_tests.txt
N=${1:-8}
for i in $(seq 1 $N); do
  flags="--no-cpu-baseline"; [ $i -eq 1 ] && flags=""
  timeout 400 python bench.py $flags > $OUT/bench_$i.json 2> $OUT/bench_$i.err; rc=$?
  python - $OUT/bench_$i.json $rc <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("bench rc=%s value %.2f M hashes/s, attempts %s, one proof %.1f ms, bn20 %.1f M / %.1f ms, gmimc %.1f M, layer_check_failures %s" % (
        sys.argv[2], d["value"] / 1e6, d.get("bench_attempts", 1), d["single_proof_latency_ms"], d["configs"]["bn20"]["hashes_per_s"] / 1e6,
        d["configs"]["bn20"]["single_proof_ms"], d["configs"]["gmimc_bn22"]["hashes_per_s"] / 1e6, d["integrity"]["layer_check_failures"]))
except Exception as e:
    print("bench rc=%s NO LINE (%s)" % (sys.argv[2], e))
P
done
