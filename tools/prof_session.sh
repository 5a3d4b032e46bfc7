#!/bin/bash
# rocprofv3 passes of a round (run on the GPU box through gpurun):  bash tools/prof_session.sh <tag>
# Kernel-trace + stats of bench.py with one proof at a time and with the default five lanes; the fold launches grouped by
# size; then the PMC passes (tools/pmc_bench.sh: FETCH_SIZE, WRITE_SIZE, SQ block -- each in its own run, --pmc with
# --kernel-trace only).  Summaries land in gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo -- python3 $ROOT/bench.py --concurrent 1 --steps 2 --warmup 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/solo.json 2> $OUT/solo.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lanes5 -- python3 $ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/lanes5.json 2> $OUT/lanes5.err
python3 - <<PY
import csv, glob, collections
for tag in ("solo", "lanes5"):
    st = glob.glob("$OUT/%s/*/*kernel_stats.csv" % tag)
    if st:
        open("$OUT/%s_kernel_stats.csv" % tag, "w").write(open(st[0]).read())
    tr = glob.glob("$OUT/%s/*/*kernel_trace.csv" % tag)
    if not tr or tag != "solo":
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        if "k_fold" in r["Kernel_Name"]:
            by[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open("$OUT/solo_fold_launches_by_size.csv", "w") as f:
        f.write('"Kernel","GridThreads","OutputsPerLaunch","Calls","AvgNs","MinNs","MaxNs","AlgorithmicBytes","GBperS_avg"\n')
        for (k, g), v in sorted(by.items(), key=lambda kv: -kv[0][1]):
            avg = sum(v) / len(v)
            f.write('"%s",%d,%d,%d,%.1f,%d,%d,%d,%.1f\n' % (k, g, g, len(v), avg, min(v), max(v), 96 * g, 96 * g / avg))
    print(open("$OUT/solo_fold_launches_by_size.csv").read()[:1500])
PY
cd $ROOT && bash tools/pmc_bench.sh 24 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_bench/summary.json $OUT/pmc_fold_traffic.json 2>/dev/null
cp gpurun_out/pmc_bench/sq_summary.json $OUT/pmc_round_kernel_sq.json 2>/dev/null
tail -3 $OUT/pmc.log
# computeH: kernel stats of three runs at 2^24 points
cat > /tmp/ch_prof.py <<PY
import importlib, sys
sys.path.insert(0, "$ROOT")
gk = importlib.import_module("gkr-mimc_amd"); gk.init(0)
ms, np_, by = gk.bench_compute_h(24, warmup=1, iters=3)
print("computeH 2^24: %.3f ms, %d passes, %.2f GB, %.0f GB/s" % (ms, np_, by / 1e9, by / ms / 1e6))
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/computeh -- python3 /tmp/ch_prof.py > $OUT/computeh.txt 2>&1
st=$(ls $OUT/computeh/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$st" ] && cp $st $OUT/computeh_2p24_kernel_stats.csv
tail -2 $OUT/computeh.txt
