#!/bin/bash
out=gpurun_out/r05_pre_start_sweep.txt
: > $out
for bn in 22 23 24; do
  for lg in 16 18 20 22 30; do
    r=$(GKRHIP_X_PRE_START_LG=$lg python tools/solo_once.py $bn 4 2>&1 | grep prove | awk '{print $2}' | sort -n | head -2 | tr '\n' ' ')
    echo "bN=$bn pre_start_lg=$lg: $r" >> $out
  done
done
r=$(GKRHIP_SPEC=2 python tools/solo_once.py 24 4 2>&1 | grep prove | awk '{print $2}' | sort -n | head -2 | tr '\n' ' '); echo "bN=24 SPEC=2: $r" >> $out
r=$(GKRHIP_SPEC=2 GKRHIP_X_PRE_START_LG=20 python tools/solo_once.py 24 4 2>&1 | grep prove | awk '{print $2}' | sort -n | head -2 | tr '\n' ' '); echo "bN=24 SPEC=2 lg=20: $r" >> $out
cat $out
