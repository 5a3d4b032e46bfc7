#!/bin/bash
# the priority/event probe in the configurations of the round-4 anomaly and its controls (profiles/r05_prio_event_probe.txt)
out=gpurun_out/r05_prio_event_probe.txt
: > $out
# built from the source in the tree every time (no committed binary: ADVICE r5)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/prio_event_probe.hip -o /tmp/prio_event_probe || exit 1
run() { echo "--- $*" >> $out; timeout 120 /tmp/prio_event_probe "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 12 25 low 16 100 17 6 1
run 12 25 normal 16 100 17 6 1
run 12 25 low 4 100 17 6 1
run 12 25 low 16 0 17 6 1
run 12 25 low 16 100 17 6 0
run 24 25 low 16 100 15 6 1
run 12 25 high 16 100 17 6 1
tail -40 $out
