B="--bn 20 --concurrent 16 --steps 64 --warmup 16 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for v in "1 1 1 1" "2 2 1 1" "2 2 2 1" "2 2 2 2" "2 0 1 1" "1 1 1 1"; do
  set -- $v
  GKRHIP_PRELAUNCH=$1 GKRHIP_SPEC=$2 GKRHIP_COOP=$3 GKRHIP_PRE=$4 GKRHIP_PRELAUNCH_LG=30 timeout 600 python bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bn20 x16 lanes: prelaunch=$1 spec=$2 coop=$3 pre=$4  %.2f M hashes/s  %.2f ms/step' % (d['value']/1e6, d['ms_per_step']))"
done
