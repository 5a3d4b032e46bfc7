#!/bin/bash
# Round kernels on few, fat lanes when many proofs are in flight (option fat_lj: 2^n pairs per lane from round 1 on): same box, interleaved.
out=gpurun_out/r06_fat_lanes_ab.txt
: > $out
run() {
  echo "--- bn=$BN lanes=$L $EXTRA $*" >> $out
  env "$@" timeout 600 python bench.py --bn $BN --concurrent $L --steps $((3*L)) --warmup $L --no-cpu-baseline --no-micro --no-oneshot --no-configs $EXTRA 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f M/s  ms_per_step %.2f single %.2f  failures %s retries %s' % (d['value']/1e6, d['ms_per_step'], d['single_proof_latency_ms'], d['integrity']['layer_check_failures'], d['integrity']['chal_retries']))
" >> $out 2>&1
}
for i in 1 2; do
  BN=20 L=24 EXTRA="" run A=1
  BN=20 L=24 EXTRA="" run GKRHIP_BENCH_OPTIONS=fat_lj=4
  BN=20 L=24 EXTRA="" run GKRHIP_BENCH_OPTIONS=fat_lj=3
  BN=20 L=24 EXTRA="" run GKRHIP_BENCH_OPTIONS=fat_lj=5
done
for i in 1 2; do
  BN=20 L=56 EXTRA="" run A=1
  BN=20 L=56 EXTRA="" run GKRHIP_BENCH_OPTIONS=fat_lj=4
done
BN=20 L=56 EXTRA="" run GKRHIP_BENCH_OPTIONS=fat_lj=5
for i in 1 2; do
  BN=22 L=12 EXTRA="--circuit gmimc" run A=1
  BN=22 L=12 EXTRA="--circuit gmimc" run GKRHIP_BENCH_OPTIONS=fat_lj=4
done
for i in 1 2; do
  BN=24 L=5 EXTRA="" run A=1
  BN=24 L=5 EXTRA="" run GKRHIP_BENCH_OPTIONS=fat_lj=4,fat_from=2
done
cat $out
