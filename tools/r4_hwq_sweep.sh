#!/bin/bash
# Throughput of BASELINE config 2 (bN = 20, 24 proofs in flight) and config 5 (GMiMC bN = 22, 12 in flight) against the number of
# hardware queues (GKRHIP_HW_QUEUES; run on the GPU box through gpurun): bash tools/r4_hwq_sweep.sh
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'wait', round(d['host_split_ms_per_step']['host_wait_ms'],1))"; }
for q in 4 6 8 10 12 16 24; do
  echo -n "bn20x24 hwq=$q: "; GKRHIP_HW_QUEUES=$q python bench.py --bn 20 --concurrent 24 --steps 48 --warmup 24 --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>/dev/null | val
done
for q in 4 8 12 16; do
  echo -n "gmimc22x12 hwq=$q: "; GKRHIP_HW_QUEUES=$q python bench.py --circuit gmimc --bn 22 --concurrent 12 --steps 24 --warmup 12 --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>/dev/null | val
done
for q in 4 8 12 16; do
  echo -n "bn24x5 hwq=$q: "; GKRHIP_HW_QUEUES=$q python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-micro --no-oneshot --no-configs 2>/dev/null | val
done
