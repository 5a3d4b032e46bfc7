#!/bin/bash
# rocprofv3 kernel trace of BASELINE config 2 with 24 proofs in flight: time and workgroup-time by kernel
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r5bn20
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bn20 -- python3 $ROOT/bench.py --bn 20 --concurrent ${1:-24} --steps 96 --warmup 24 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench.json 2>$OUT/err.txt < /dev/null
f=$(ls /tmp/bn20/*/*kernel_stats.csv 2>/dev/null | head -1); t=$(ls /tmp/bn20/*/*kernel_trace.csv 2>/dev/null | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 - "$t" "$OUT/bench.json" <<'PY' | tee $OUT/summary.txt
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print('value %.2f M/s ms_per_step %.2f' % (d['value'] / 1e6, d['ms_per_step']))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']))
    key = (r['Kernel_Name'][:64], wgs)
    a = agg[key]; a[0] += 1; a[1] += dur; a[2] += dur * min(wgs, 512)
print('span %.1f ms, %d launches' % ((t1 - t0) / 1e6, len(rows)))
tot = sum(a[2] for a in agg.values())
print('%-66s %7s %7s %9s %9s %6s' % ('kernel', 'wgs', 'calls', 'avg us', 'sum ms', 'wg-t%'))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    print('%-66s %7d %7d %9.1f %9.2f %6.1f' % (k[0], k[1], a[0], a[1] / a[0], a[1] / 1e3, 100 * a[2] / tot))
PY
