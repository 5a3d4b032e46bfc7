#!/bin/bash
# Round-4 A/B of the runtime's hardware-queue count on ONE box, interleaved: the default (4) against GKRHIP_HW_QUEUES=16 for
# (1) bN = 24 with 5 lanes, which also reports the proof alone on the GPU (median of five), (2) bN = 20 with 24 lanes,
# (3) GMiMC bN = 22 with 12 lanes.  Usage: bash tools/r4_hwq_ab.sh <tag> [samples]
TAG=${1:-r4_hwq}; N=${2:-5}
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-micro --no-oneshot --no-configs"
export GKRHIP_BENCH_CHILD=1      # no child-process re-measurement inside bench.py
one() {  # name hwq args...
  name=$1; hq=$2; shift 2
  f=$OUT/${name}_hwq${hq}_$RANDOM
  if [ "$hq" = "0" ]; then timeout 600 python bench.py $B "$@" > $f.json 2> $f.err
  else GKRHIP_HW_QUEUES=$hq timeout 600 python bench.py $B "$@" > $f.json 2> $f.err; fi
  python3 - $f.json $name $hq <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s hwq=%s value %.2f M/s solo %.1f ms sclk %s" % (sys.argv[2], sys.argv[3], d["value"] / 1e6, d["config"]["single_proof_latency_ms"],
          d.get("sclk_mhz_during_timed_steps")))
except Exception as e:
    print(sys.argv[1], "no json:", e)
PY
}
for i in $(seq $N); do
  for hq in 0 16; do
    one bn24x5 $hq
    one bn20x24 $hq --bn 20 --concurrent 24 --steps 48 --warmup 24
    one gmimc22x12 $hq --circuit gmimc --bn 22 --concurrent 12 --steps 24 --warmup 12
  done
done
python3 - $OUT <<'PY'
import glob, json, os, statistics, sys
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    b = os.path.basename(f)
    if b == "summary.json":
        continue
    name, hq = b.split("_hwq")[0], b.split("_hwq")[1].split("_")[0]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows.setdefault((name, hq), []).append((d["value"], d["config"]["single_proof_latency_ms"]))
out = {}
for (name, hq), v in sorted(rows.items()):
    out["%s hwq=%s" % (name, hq)] = {"n": len(v), "value_M_per_s": [round(x[0] / 1e6, 2) for x in v], "median_value_M_per_s": round(statistics.median(x[0] for x in v) / 1e6, 2),
                                     "solo_ms": [round(x[1], 1) for x in v], "median_solo_ms": round(statistics.median(x[1] for x in v), 1)}
json.dump(out, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
