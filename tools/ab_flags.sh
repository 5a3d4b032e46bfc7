#!/bin/bash
# A/B of two builds on ONE box: default flags, then GKRHIP_EXTRA_FLAGS="$2" (rebuilt on the box), each benched twice.
# Usage: bash tools/ab_flags.sh <tag> "<extra flags>"
TAG=$1; FL=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-micro --no-oneshot --no-configs ${AB_BENCH_ARGS:-}"
run() { timeout 600 python bench.py $B > $OUT/$1.json 2> $OUT/$1.err; python3 - $OUT/$1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pe = d["partial_eval"]
print(sys.argv[1], round(d["value"] / 1e6, 2), "M/s; round0", round(pe["avg_launch_ms"], 4), "ms; vector", pe["loop_instructions_per_pair"]["vector"], "latency", d["config"]["single_proof_latency_ms"])
PY
}
run a1
GKRHIP_EXTRA_FLAGS="$FL" python -c "import importlib; importlib.import_module('gkr-mimc_amd.build').build(force=True)" > $OUT/build_b.log 2>&1
export GKRHIP_EXTRA_FLAGS="$FL"
run b1
unset GKRHIP_EXTRA_FLAGS
python -c "import importlib; importlib.import_module('gkr-mimc_amd.build').build(force=True)" > $OUT/build_a.log 2>&1
run a2
GKRHIP_EXTRA_FLAGS="$FL" python -c "import importlib; importlib.import_module('gkr-mimc_amd.build').build(force=True)" >> $OUT/build_b.log 2>&1
export GKRHIP_EXTRA_FLAGS="$FL"
run b2
