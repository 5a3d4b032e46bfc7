#!/bin/bash
# HBM traffic of the fold launches INSIDE bench.py's workload, from the PMC counters collected as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, --pmc together
# with --kernel-trace only).  Run on the GPU box from the repo root:  bash tools/pmc_bench.sh [bn]
# Writes gpurun_out/pmc_bench/summary.json (copy it to profiles/r01_pmc_fold_traffic.json).
BN=${1:-24}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $ROOT/bench.py --bn $BN --steps 2 --warmup 0 --concurrent 1 --no-cpu-baseline > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, json
res = {"bn": $BN, "command": "bench.py --bn $BN --steps 2 --warmup 0 --concurrent 1 --no-cpu-baseline"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % c)
    vals = []
    for r in csv.DictReader(open(fs[0])):
        if r.get("Kernel_Name", "").startswith("void k_fold<2>") and r.get("Counter_Name") == c:
            vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    big = [v for g, v in vals if g == max(g for g, _ in vals)]
    top = sorted(big)[-3:]            # the round-0 launches (largest tables) of the proofs in the run
    res[c + "_KB_per_round0_fold_launch"] = sum(top) / len(top)
    res[c + "_dispatches_seen"] = len(vals)
# gfx950: FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM)
res["traffic_bytes_per_launch"] = (2 * res["FETCH_SIZE_KB_per_round0_fold_launch"] + res["WRITE_SIZE_KB_per_round0_fold_launch"]) * 1024
res["algorithmic_bytes_per_launch"] = 96 * 2 * (1 << ($BN - 1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
