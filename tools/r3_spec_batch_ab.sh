#!/bin/bash
# same-box A/B: all speculative launches of a layer queued at once (GKRHIP_SPEC_BATCH=1) or one per round
B="--concurrent 1 --steps 6 --warmup 2 --no-cpu-baseline --no-micro --no-oneshot --no-configs"
for bn in ${@:-20}; do
for v in 0 1 0 1 0 1; do
  GKRHIP_SPEC_BATCH=$v timeout 600 python bench.py --bn $bn $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sp=d.get('single_proof',{})
print('bn=$bn batch=$v latency %.1f ms (samples %s) hash %.1f wait %.1f launch %.1f other %.1f' % (d['config']['single_proof_latency_ms'], ' '.join('%.1f' % x for x in sp.get('latency_samples_ms',[])), sp.get('host_hash_ms',0), sp.get('host_wait_ms',0), sp.get('host_launch_ms',0), sp.get('host_other_ms',0)))"
done; done
