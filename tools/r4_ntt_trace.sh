cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ntt_trace -- python3 $GRAFT_REPO_ROOT/tools/computeh_bench.py 24 --iters 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ntt_trace/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_ntt_tile" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-9:]
names = ["DIF s7 x3", "DIF s6 x3", "DIF c11 x3", "DIT c11 pre x3", "DIT s6 x3", "DIT s7 x3", "DIF s7 pointwise", "DIF s6 x1", "DIF c11 post x1"]
tot = 0
for n, r in zip(names, last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += d
    print("%-18s %s grid %s x %s  %.3f ms" % (n, r["Kernel_Name"][:22], r["Grid_Size_X"], r["Grid_Size_Y"], d))
print("total", tot)
PY
