#!/bin/bash
out=gpurun_out/r05_knobs20.txt
: > $out
run() { r=$(env "$@" python tools/solo_once.py 20 5 2>&1 | grep prove | awk '{print $2}' | sort -n | head -3 | tr '\n' ' '); echo "$*: $r" >> $out; }
run A=0
run GKRHIP_COOP_LG=15
run GKRHIP_COOP_LG=16
run GKRHIP_COOP_LG=13
run GKRHIP_SPEC_LG=14
run GKRHIP_SPEC_LG=12
run GKRHIP_HOST_TAIL=3
run GKRHIP_HOST_TAIL=5
run GKRHIP_WT_LATE_LJ=1
run GKRHIP_WT_LATE_LJ=2
run GKRHIP_PRE=0
cat $out
