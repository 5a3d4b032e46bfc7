#!/bin/bash
# kernel launches per gkr.Prove at bN = 20: the difference of two rocprofv3 runs with 2 and 6 proofs, divided by 4
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for p in 2 6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lc_$p -- python3 $ROOT/tools/launch_count.py ${1:-20} $p > /dev/null 2>&1 < /dev/null
  f=$(ls /tmp/lc_$p/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 -c "
import csv,sys
rows=list(csv.DictReader(open('$f')))
print('$p proofs: launches', sum(int(r['Calls']) for r in rows), 'kernel time ms', round(sum(int(r['TotalDurationNs']) for r in rows)/1e6,2))
for r in rows[:12]: print('   ', r['Name'][:70], r['Calls'], round(int(r['TotalDurationNs'])/1e6,2))
"
done
