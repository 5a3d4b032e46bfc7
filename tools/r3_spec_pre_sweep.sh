B="--concurrent 1 --steps 4 --warmup 2 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for bn in 24 23; do
for v in "0 16" "2 16" "2 17" "2 18" "2 19" "0 16"; do
  set -- $v
  GKRHIP_SPEC=$1 GKRHIP_PRE_START_LG=$2 timeout 600 python bench.py --bn $bn $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sp=d.get('single_proof',{})
print('bn=$bn spec=$1 pre_start=$2 latency %.1f ms hash %.1f wait %.1f launch %.1f other %.1f' % (d['config']['single_proof_latency_ms'], sp.get('host_hash_ms',0), sp.get('host_wait_ms',0), sp.get('host_launch_ms',0), sp.get('host_other_ms',0)))"
done; done
