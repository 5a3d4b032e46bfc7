#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 5 (GMiMC bN = 22) with 12 proofs in flight
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/r4gm
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gm -- python3 $ROOT/bench.py --circuit gmimc --bn 22 --concurrent 12 --steps 36 --warmup 12 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $ROOT/gpurun_out/r4gm/bench.json 2>/dev/null < /dev/null
f=$(ls /tmp/gm/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $ROOT/gpurun_out/r4gm/gmimc22_lanes12_kernel_stats.csv && python3 -c "
import csv,json
rows=list(csv.DictReader(open('$f')))
d=json.loads(open('$ROOT/gpurun_out/r4gm/bench.json').read().strip().splitlines()[-1])
print('value', d['value']/1e6, 'ms_per_step', d['ms_per_step'])
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('launches', sum(int(r['Calls']) for r in rows), 'kernel time ms', round(tot/1e6,2))
for r in rows[:16]: print('   ', r['Name'][:72], r['Calls'], round(int(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1), r['Percentage'])
"
