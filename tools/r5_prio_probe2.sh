#!/bin/bash
# second set: what about the re-recorded event, the store flavour, an explicit release, and what the runtime believes
out=gpurun_out/r05_prio_event_probe2.txt
: > $out
run() { echo "--- $*" >> $out; timeout 150 tools/prio_event_probe_bin "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 12 30 low 16 100 17 6 1 0 0 1
run 12 30 low 16 100 17 6 2 0 0 0
run 12 30 low 16 100 17 6 1 1 0 0
run 12 30 low 16 100 17 6 1 0 1 0
run 12 40 low 16 100 17 6 0 0 0 0
run 12 40 normal 16 100 17 6 1 0 0 0
grep -v "^lane" $out | tail -40
