#!/bin/bash
out=gpurun_out/r05_prio_rounds2.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $EXTRA $*" >> $out
  env "$@" timeout 400 python bench.py --bn $BN --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs $EXTRA 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f' % (d['value']/1e6, d['ms_per_step']))
" >> $out 2>&1
}
for i in 1 2; do for m in 0123 0233 0333 0112 0223; do BN=24 L=5 EXTRA="" run GKRHIP_X_ROUND_PRIO=$m; done; done
for i in 1 2; do for m in 0123 0233 0333 0112 0223; do BN=22 L=12 EXTRA="--circuit gmimc" run GKRHIP_X_ROUND_PRIO=$m; done; done
for m in 0123 0233 0333 0112 0223; do BN=20 L=24 EXTRA="" run GKRHIP_X_ROUND_PRIO=$m; done
cat $out
