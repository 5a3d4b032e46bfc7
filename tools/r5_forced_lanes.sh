#!/bin/bash
# bN = 20: lanes with the solo paths forced on (pre-launched rounds, speculation, cooperative kernel, look-ahead), throughput
out=gpurun_out/r05_forced_lanes.txt
: > $out
run() {
  echo "--- $*" >> $out
  env "$@" timeout 300 python bench.py --bn 20 --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
print('host split', d.get('host_split_ms_per_step'))
" >> $out 2>&1
}
L=24 run GKRHIP_HOST_TAIL=5
L=12 run GKRHIP_PRELAUNCH=2 GKRHIP_SPEC=2 GKRHIP_COOP=2
L=24 run GKRHIP_PRELAUNCH=2 GKRHIP_SPEC=2 GKRHIP_COOP=2
L=24 run GKRHIP_PRELAUNCH=2 GKRHIP_SPEC=2 GKRHIP_COOP=0
L=24 run GKRHIP_PRELAUNCH=2 GKRHIP_SPEC=0 GKRHIP_COOP=0
L=24 run GKRHIP_PRELAUNCH=2 GKRHIP_SPEC=2 GKRHIP_COOP=2 GKRHIP_PRE=2
L=32 run GKRHIP_PRELAUNCH=2 GKRHIP_SPEC=2 GKRHIP_COOP=0
cat $out
