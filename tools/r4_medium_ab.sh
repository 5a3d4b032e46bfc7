#!/bin/bash
# Round-4 A/B of the medium rounds (2^14..2^16 pairs) of a proof that is alone on the GPU, same box, interleaved:
# cooperative kernel up to 2^15 / 2^16 pairs (GKRHIP_COOP_LG), speculative rounds up to 2^14 (GKRHIP_SPEC_LG).
# Usage: bash tools/r4_medium_ab.sh <tag> [samples] [bn...]
TAG=${1:-r4_medium}; N=${2:-3}; shift 2
BNS="${@:-20 22}"
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--concurrent 1 --steps 4 --warmup 2 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for i in $(seq $N); do
 for bn in $BNS; do
  for v in "14 13" "15 13" "16 13" "14 14" "16 14"; do
    co=${v% *}; sp=${v#* }
    f=$OUT/bn${bn}_coop${co}_spec${sp}_$RANDOM
    GKRHIP_COOP_LG=$co GKRHIP_SPEC_LG=$sp timeout 600 python bench.py --bn $bn $B > $f.json 2> $f.err
  done
 done
done
python3 - $OUT <<'PY'
import glob, json, os, statistics, sys
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bn*.json"))):
    key = "_".join(os.path.basename(f).split("_")[:3])
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.setdefault(key, []).append(d["config"]["single_proof_latency_ms"])
    except Exception as e:
        print(f, "no json", e)
out = {k: {"samples_ms": [round(x, 2) for x in v], "median_ms": round(statistics.median(v), 2)} for k, v in sorted(rows.items())}
json.dump(out, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
