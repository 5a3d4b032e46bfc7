#!/bin/bash
# host-tail depth sweep at bN = 20: 24 lanes (throughput) and the proof alone
out=gpurun_out/r05_host_tail_sweep.txt
: > $out
for h in ${@:-5 7 8 9 10}; do
  echo "--- GKRHIP_HOST_TAIL=$h" >> $out
  GKRHIP_HOST_TAIL=$h timeout 300 python bench.py --bn 20 --concurrent 24 --steps 72 --warmup 24 --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f  single %.2f ms' % (d['value']/1e6, d['ms_per_step'], d.get('single_proof_latency_ms',0)))
print('host split', d.get('host_split_ms_per_step'))
" >> $out 2>&1
done
cat $out
