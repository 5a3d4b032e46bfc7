#!/bin/bash
# HBM traffic of the fold launches INSIDE bench.py's workload, from the PMC counters collected as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, --pmc together
# with --kernel-trace only).  Run on the GPU box from the repo root:  bash tools/pmc_bench.sh [bn]
# Writes gpurun_out/pmc_bench/summary.json (copy it to profiles/r02_pmc_fold_traffic.json).
BN=${1:-24}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $ROOT/bench.py --bn $BN --steps 2 --warmup 0 --concurrent 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, json
res = {"bn": $BN, "command": "bench.py --bn $BN --steps 2 --warmup 0 --concurrent 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % c)
    vals = []
    for r in csv.DictReader(open(fs[0])):
        if r.get("Kernel_Name", "").startswith("void k_fold<") and r.get("Counter_Name") == c:
            vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    big = [v for g, v in vals if g == max(g for g, _ in vals)]
    # the launches on the largest tables (round 0 of the key-copy layer: one single-table launch per table of the
    # instance, plus bench.py's 20-launch roofline loop on a table of the same size): average per launch
    res[c + "_KB_per_round0_fold_launch"] = sum(big) / len(big)
    res[c + "_dispatches_seen"] = len(vals)
# gfx950: FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM)
res["traffic_bytes_per_launch"] = (2 * res["FETCH_SIZE_KB_per_round0_fold_launch"] + res["WRITE_SIZE_KB_per_round0_fold_launch"]) * 1024
res["algorithmic_bytes_per_launch"] = 96 * (1 << ($BN - 1))
res["note"] = "k_fold<1>: one launch per table of 2^bn elements; per launch, as bench.py's roofline.algorithmic_bytes_per_launch"
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res))
PY

# VALU-side counters of the round kernels (one more pass, SQ block only)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/SQ -- python3 $ROOT/bench.py --bn $BN --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/SQ.log 2>&1
python3 - <<PY
import csv, glob, json, collections
fs = glob.glob("$OUT/SQ/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r.get("Kernel_Name", "")
    if "k_cipher_round" in k or "k_fold" in k:
        key = (k.split("(")[0], int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[key] += 1
rows = []
for key in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:8]:
    a = agg[key]
    rows.append({"kernel": key[0], "grid_threads": key[1], "dispatches": cnt[key], **{c: a[c] for c in sorted(a)},
                 "valu_active_fraction_of_wave_cycles": a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else None,
                 "valu_insts_per_wave": a["SQ_INSTS_VALU"] / a["SQ_WAVES"] if a.get("SQ_WAVES") else None})
json.dump({"bn": $BN, "command": "bench.py --bn $BN --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs", "kernels": rows}, open("$OUT/sq_summary.json", "w"), indent=1)
print(json.dumps(rows[:3]))
PY
