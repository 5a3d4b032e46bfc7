#!/bin/bash
# Do the lanes' co-resident kernel VARIANTS cost (instruction cache: a wide round kernel's loop is ~40 KB of code, the cache 64 KB per
# two CUs)?  Fewer distinct variants in flight, same arithmetic: bN = 20 x 24 lanes, same box, interleaved.
out=gpurun_out/r06_variants_ab.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $*" >> $out
  env "$@" timeout 600 python bench.py --bn $BN --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f single %.2f' % (d['value']/1e6, d['ms_per_step'], d['single_proof_latency_ms']))
" >> $out 2>&1
}
for i in 1 2; do
  BN=20 L=24 run A=1
  BN=20 L=24 run GKRHIP_BENCH_OPTIONS=wt_late_lj=99
  BN=20 L=24 run GKRHIP_LAT=0
  BN=20 L=24 run GKRHIP_AHEAD=0
  BN=20 L=24 run GKRHIP_BENCH_OPTIONS=wt_late_lj=99 GKRHIP_LAT=0 GKRHIP_AHEAD=0
  BN=20 L=24 run GKRHIP_WIDE=0
done
cat $out
