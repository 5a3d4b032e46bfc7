#!/bin/bash
# SQ-block counters of the MSM and computeH kernels (one rocprofv3 pass each: --pmc with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes).  Run on the GPU box:  bash tools/r4_pmc_msm_ntt.sh
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/msm -- python3 $ROOT/tools/msm_bench.py 22 > $OUT/msm.log 2>&1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/ntt -- python3 $ROOT/tools/computeh_bench.py 24 --iters 1 > $OUT/ntt.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, json, collections
out = {}
for tag, pats in (("msm", ("k_msm_",)), ("ntt", ("k_ntt_tile",))):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % tag)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r.get("Kernel_Name", "")
        if any(p in k for p in pats):
            key = k.split("(")[0]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": cnt[key] += 1
    rows = []
    for key in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:6]:
        a = agg[key]
        rows.append({"kernel": key, "dispatches": cnt[key], **{c: a[c] for c in sorted(a)},
                     "valu_active_fraction_of_wave_cycles": a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else None,
                     "valu_insts_per_wave": a["SQ_INSTS_VALU"] / a["SQ_WAVES"] if a.get("SQ_WAVES") else None})
    out[tag] = rows
json.dump({"commands": ["tools/msm_bench.py 22", "tools/computeh_bench.py 24 --iters 1"], "counters": "$C", "kernels": out},
          open("$OUT/sq_summary.json", "w"), indent=1)
for tag in out:
    for r in out[tag][:3]:
        print(tag, r["kernel"], r["dispatches"], "valu_active/wave_cycles", r["valu_active_fraction_of_wave_cycles"], "valu insts/wave", r["valu_insts_per_wave"])
PY
rm -rf $OUT/msm $OUT/ntt
