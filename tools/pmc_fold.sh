#!/bin/bash
# HBM traffic of the fold kernel from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes, --pmc together with --kernel-trace only.
# Run on the GPU box from the repo root:  bash tools/pmc_fold.sh [bn]
BN=${1:-24}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_fold
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $ROOT/tools/fold_only.py $BN > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % c)
    if not fs:
        print("no counter file for", c); continue
    vals = []
    for r in csv.DictReader(open(fs[0])):
        if r.get("Kernel_Name", "").startswith("k_fold") and r.get("Counter_Name") == c:
            vals.append(float(r["Counter_Value"]))
    res[c] = vals
    print(c, "k_fold dispatches:", len(vals), "max:", max(vals) if vals else None, "mean:", sum(vals)/len(vals) if vals else None)
PY
