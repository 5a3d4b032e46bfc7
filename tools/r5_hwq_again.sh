#!/bin/bash
out=gpurun_out/r05_hwq_with_priorities.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $EXTRA $*" >> $out
  env "$@" timeout 600 python bench.py --bn $BN --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs $EXTRA 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
" >> $out 2>&1
}
for q in 8 12 16 24; do BN=24 L=5 EXTRA="" run GPU_MAX_HW_QUEUES=$q; done
for q in 8 12 16 24; do BN=22 L=12 EXTRA="--circuit gmimc" run GPU_MAX_HW_QUEUES=$q; done
for q in 8 12 16; do BN=20 L=56 EXTRA="" run GPU_MAX_HW_QUEUES=$q; done
cat $out
