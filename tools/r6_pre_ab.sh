#!/bin/bash
# Does the look-ahead kernel (k_cipher_pre: the q-independent products of the next layer's round 0, GKRHIP_PRE) still earn its
# keep now that round 0 runs ahead of its point (GKRHIP_AHEAD)?  One proof alone, same box, interleaved.
out=gpurun_out/r06_pre_ab.txt
: > $out
run() { echo "--- bn=$BN $*" >> $out; env "$@" python tools/solo_once.py $BN 5 2>&1 | tr '\n' ' ' >> $out; echo >> $out; }
for i in 1 2; do
  for BN in 22 23 24; do
    BN=$BN run A=1
    BN=$BN run GKRHIP_PRE=0
  done
done
BN=24 run GKRHIP_PRE=0 GKRHIP_SPEC=2
BN=25 run A=1
BN=25 run GKRHIP_PRE=0
cat $out
