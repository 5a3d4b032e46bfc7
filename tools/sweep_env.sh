#!/bin/bash
# Same-box sweep of run-time switches (environment) over bench.py's default workload.
# Usage: bash tools/sweep_env.sh <tag> "VAR=a VAR=b ..."   (each item is one run; "-" = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for item in "$@"; do
  name=$(echo "$item" | tr ' =' '__')
  if [ "$item" = "-" ]; then env_prefix=""; else env_prefix="$item"; fi
  env $env_prefix timeout 600 python bench.py --no-cpu-baseline --no-micro --no-oneshot ${SWEEP_BENCH_ARGS:-} > $OUT/$name.json 2> $OUT/$name.err
  python3 - $OUT/$name.json "$item" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-40s %6.2f M/s  single %6.1f ms  round0 %.4f ms" % (sys.argv[2], d["value"] / 1e6, d["config"]["single_proof_latency_ms"], d["partial_eval"]["avg_launch_ms"]))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
