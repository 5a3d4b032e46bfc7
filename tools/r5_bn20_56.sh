#!/bin/bash
out=gpurun_out/r05_bn20_56lanes_knobs.txt
: > $out
run() {
  echo "--- bn=20 lanes=$L $*" >> $out
  env "$@" timeout 600 python bench.py --bn 20 --concurrent $L --steps $((2*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
" >> $out 2>&1
}
L=56 run A=1
L=56 run GKRHIP_HOST_TAIL=6
L=56 run GKRHIP_HOST_TAIL=4
L=56 run GKRHIP_GMAX=14
L=56 run GPU_MAX_HW_QUEUES=24
L=56 run GKRHIP_WAIT_SPIN_US=5
L=56 run GKRHIP_WAIT_SPIN_US=100
L=56 run A=2
L=72 run A=1
cat $out
