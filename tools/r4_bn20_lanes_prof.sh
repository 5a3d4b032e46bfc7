#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 2 with 24 proofs in flight: launches and kernel time per proof
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/r4bn20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bn20 -- python3 $ROOT/bench.py --bn 20 --concurrent 24 --steps 96 --warmup 24 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $ROOT/gpurun_out/r4bn20/bench.json 2>/dev/null < /dev/null
f=$(ls /tmp/bn20/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $ROOT/gpurun_out/r4bn20/bn20_lanes24_kernel_stats.csv && python3 -c "
import csv,json
rows=list(csv.DictReader(open('$f')))
d=json.loads(open('$ROOT/gpurun_out/r4bn20/bench.json').read().strip().splitlines()[-1])
print('value', d['value']/1e6, 'ms_per_step', d['ms_per_step'])
print('launches', sum(int(r['Calls']) for r in rows), 'kernel time ms', round(sum(int(r['TotalDurationNs']) for r in rows)/1e6,2))
for r in rows[:14]: print('   ', r['Name'][:70], r['Calls'], round(int(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
"
