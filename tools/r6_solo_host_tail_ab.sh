out=gpurun_out/r06_solo_host_tail_ab.txt
: > $out
run() { echo "--- bn=$BN $*" >> $out; env "$@" python tools/solo_once.py $BN 5 2>&1 | tr '\n' ' ' >> $out; echo >> $out; }
for i in 1 2; do for BN in 24 22 20; do BN=$BN run A=1; BN=$BN run GKRHIP_HOST_TAIL=5; BN=$BN run GKRHIP_HOST_TAIL=3; done; done
cat $out
