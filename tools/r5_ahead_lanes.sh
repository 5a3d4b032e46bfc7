#!/bin/bash
# round 0 ahead of its point forced for every lane (GKRHIP_AHEAD=2) against the default (only a proof alone): throughput
out=gpurun_out/r05_ahead_lanes.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $*" >> $out
  env "$@" timeout 400 python bench.py --bn $BN --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs $EXTRA 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
" >> $out 2>&1
}
BN=20 L=24 run GKRHIP_AHEAD=1
BN=20 L=24 run GKRHIP_AHEAD=2
BN=20 L=24 run GKRHIP_AHEAD=2 GKRHIP_HOST_TAIL=7
BN=20 L=16 run GKRHIP_AHEAD=2
BN=24 L=5 run GKRHIP_AHEAD=1
BN=24 L=5 run GKRHIP_AHEAD=2
BN=22 L=12 EXTRA="--circuit gmimc" run GKRHIP_AHEAD=1
BN=22 L=12 EXTRA="--circuit gmimc" run GKRHIP_AHEAD=2
cat $out
