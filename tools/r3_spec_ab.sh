#!/bin/bash
# A/B of the speculative small rounds on ONE box (a proof alone on the GPU).  Usage: bash tools/r3_spec_ab.sh <tag> [bn...]
TAG=${1:-specab}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--concurrent 1 --steps 4 --warmup 2 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for bn in ${@:-20 24}; do
  for v in "0 13" "1 12" "1 13" "1 14" "0 13" "1 13"; do
    set -- $v
    f=$OUT/solo_bn${bn}_spec$1_lg$2_$RANDOM
    GKRHIP_SPEC=$1 GKRHIP_SPEC_LG=$2 timeout 600 python bench.py --bn $bn $B > $f.json 2> $f.err
    python3 - $f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sp = d.get("single_proof", {})
    print(sys.argv[1], "latency %.1f ms" % d["config"]["single_proof_latency_ms"],
          "hash %.1f wait %.1f launch %.1f other %.1f" % tuple(sp.get(k, 0) for k in ("host_hash_ms", "host_wait_ms", "host_launch_ms", "host_other_ms")))
except Exception as e:
    print(sys.argv[1], "no json:", e)
PY
  done
done
