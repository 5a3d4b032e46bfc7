#!/bin/bash
# When the look-ahead kernel of the next layer is queued (option pre_start_lg: at the round with 2^n pairs): one proof alone, same box.
out=gpurun_out/r06_pre_start_ab.txt
: > $out
run() { echo "--- bn=$BN pre_start_lg=$1" >> $out; GKR_SOLO_OPTIONS=pre_start_lg=$1 python tools/solo_once.py $BN 5 2>&1 | tr '\n' ' ' >> $out; echo >> $out; }
for i in 1 2; do for BN in 24 23; do for v in 20 21 22 19; do BN=$BN run $v; done; done; done
cat $out
