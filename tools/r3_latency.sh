#!/bin/bash
# Round 3: A/B of the serial-latency switches on ONE box (a proof alone on the GPU).  Usage: bash tools/r3_latency.sh <tag> [bn...]
TAG=${1:-lat}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--concurrent 1 --steps 4 --warmup 2 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for bn in ${@:-24 20}; do
  for v in "0 0 0" "1 0 0" "0 1 0" "1 1 0" "1 1 1"; do
    set -- $v
    f=$OUT/solo_bn${bn}_pl$1_pre$2_coop$3
    GKRHIP_PRELAUNCH=$1 GKRHIP_PRE=$2 GKRHIP_COOP=$3 timeout 600 python bench.py --bn $bn $B > $f.json 2> $f.err
    python3 - $f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sp = d.get("single_proof", {})
    print(sys.argv[1], "latency %.1f ms" % d["config"]["single_proof_latency_ms"], "step %.1f ms" % d["ms_per_step"],
          "hash %.1f wait %.1f launch %.1f other %.1f" % tuple(sp.get(k, 0) for k in ("host_hash_ms", "host_wait_ms", "host_launch_ms", "host_other_ms")),
          "round0 %.3f ms" % d.get("partial_eval", {}).get("avg_launch_ms", 0))
except Exception as e:
    print(sys.argv[1], "no json:", e)
PY
  done
done
