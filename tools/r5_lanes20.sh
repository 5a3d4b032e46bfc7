#!/bin/bash
out=gpurun_out/r05_lanes20.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $*" >> $out
  env "$@" timeout 600 python bench.py --bn $BN --concurrent $L --steps $((2*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
" >> $out 2>&1
}
for l in 24 40 48 56 64 80; do BN=20 L=$l run A=1; done
BN=20 L=48 run GPU_MAX_HW_QUEUES=24
BN=20 L=48 run GPU_MAX_HW_QUEUES=32
BN=20 L=64 run GPU_MAX_HW_QUEUES=32
cat $out
