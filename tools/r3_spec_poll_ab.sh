#!/bin/bash
# same-box A/B: speculative launches polling for their challenge (GKRHIP_SPEC_POLL=1) or launched with it as an argument (0)
B="--concurrent 1 --steps 4 --warmup 2 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for bn in ${@:-20 22 23 24}; do
for v in "1 1" "2 0" "2 1" "2 0" "0 1"; do
  set -- $v
  GKRHIP_SPEC=$1 GKRHIP_SPEC_POLL=$2 timeout 600 python bench.py --bn $bn $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sp=d.get('single_proof',{})
print('bn=$bn spec=$1 poll=$2 latency %.1f ms hash %.1f wait %.1f launch %.1f other %.1f spec_rounds %d' % (d['config']['single_proof_latency_ms'], sp.get('host_hash_ms',0), sp.get('host_wait_ms',0), sp.get('host_launch_ms',0), sp.get('host_other_ms',0), sp.get('spec_rounds',0)))"
done; done
