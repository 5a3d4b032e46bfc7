#!/bin/bash
# second set: what about the re-recorded event, the store flavour, an explicit release, and what the runtime believes
out=gpurun_out/r05_prio_event_probe2.txt
: > $out
# built from the source in the tree every time (no committed binary: ADVICE r5)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/prio_event_probe.hip -o /tmp/prio_event_probe || exit 1
run() { echo "--- $*" >> $out; timeout 150 /tmp/prio_event_probe "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 12 30 low 16 100 17 6 1 0 0 1
run 12 30 low 16 100 17 6 2 0 0 0
run 12 30 low 16 100 17 6 1 1 0 0
run 12 30 low 16 100 17 6 1 0 1 0
run 12 40 low 16 100 17 6 0 0 0 0
run 12 40 normal 16 100 17 6 1 0 0 0
grep -v "^lane" $out | tail -40
