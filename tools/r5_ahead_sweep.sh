#!/bin/bash
# one proof alone: round 0 ahead (GKRHIP_AHEAD) x look-ahead products (GKRHIP_PRE), by size
out=gpurun_out/r05_ahead_sweep.txt
: > $out
for bn in ${@:-18 20 21 22 23 24}; do
  for cfg in "1 1" "1 0" "0 1" "0 0"; do
    set -- $cfg
    r=$(GKRHIP_AHEAD=$1 GKRHIP_PRE=$2 python tools/solo_once.py $bn 4 2>&1 | grep prove | awk '{print $2}' | sort -n | head -2 | tr '\n' ' ')
    echo "bN=$bn AHEAD=$1 PRE=$2: $r" >> $out
  done
done
cat $out
