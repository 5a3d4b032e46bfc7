#!/bin/bash
# with the round kernels' wave priorities in: the thread cap and the number of lanes again
out=gpurun_out/r05_after_prio_sweep.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $EXTRA $*" >> $out
  env "$@" timeout 400 python bench.py --bn $BN --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs $EXTRA 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
" >> $out 2>&1
}
for g in 14 15 16; do BN=20 L=24 EXTRA="" run GKRHIP_GMAX=$g; done
for l in 16 32 48; do BN=20 L=$l EXTRA="" run A=1; done
for g in 15 16 17; do BN=24 L=5 EXTRA="" run GKRHIP_GMAX=$g; done
for g in 15 16; do BN=22 L=12 EXTRA="--circuit gmimc" run GKRHIP_GMAX=$g; done
for l in 8 16; do BN=22 L=$l EXTRA="--circuit gmimc" run A=1; done
cat $out
