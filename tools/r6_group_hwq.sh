#!/bin/bash
# Proof groups against the number of hardware queues (GPU_MAX_HW_QUEUES; the library asks for 16): 24 groups of 3 share 16 queues.
cd ${GRAFT_REPO_ROOT:-$PWD}
for q in 16 24 32; do
  echo "--- GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 400 python tools/r6_group_probe.py 20 72 3,4 3 2>&1 | grep "in flight"
done
echo "--- 64 in flight in groups of 4: sixteen groups, one per queue (16 queues)"
timeout 400 python tools/r6_group_probe.py 20 64 4 3 2>&1 | grep "in flight"
echo "--- 48 in flight in groups of 3: sixteen groups (16 queues)"
timeout 400 python tools/r6_group_probe.py 20 48 3 3 2>&1 | grep "in flight"
