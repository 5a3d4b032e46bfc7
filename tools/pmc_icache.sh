#!/bin/bash
# Instruction-cache counters of the round kernels (one rocprofv3 --pmc pass with --kernel-trace only), one proof at a
# time and with five proofs in flight.  Run on the GPU box from the repo root:  bash tools/pmc_icache.sh <tag>
TAG=${1:-icache}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/solo -- python3 $ROOT/bench.py --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline --no-micro --no-oneshot > $OUT/solo.log 2>&1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/lanes5 -- python3 $ROOT/bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-micro --no-oneshot > $OUT/lanes5.log 2>&1
python3 - <<PY
import csv, glob, json, collections
res = {}
for tag in ("solo", "lanes5"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % tag)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_IFETCH":
            n[k] += 1
    out = []
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
        d = dict(c)
        d["kernel"] = k
        d["dispatches"] = n[k]
        if d.get("SQC_ICACHE_REQ"):
            d["icache_miss_rate"] = d.get("SQC_ICACHE_MISSES", 0) / d["SQC_ICACHE_REQ"]
        if d.get("SQ_WAVE_CYCLES"):
            d["wait_inst_any_fraction_of_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
        out.append(d)
    res[tag] = out
json.dump(res, open("$OUT/icache_summary.json", "w"), indent=1)
for tag, out in res.items():
    for d in out[:5]:
        print(tag, d["kernel"], "miss rate", round(d.get("icache_miss_rate", -1), 4), "req", d.get("SQC_ICACHE_REQ"), "ifetch", d.get("SQ_IFETCH"),
              "wait_inst_any", round(d.get("wait_inst_any_fraction_of_wave_cycles", -1), 3))
PY
