#!/bin/bash
# Soaks of the round-6 tree (run on the GPU box through gpurun): every proof byte-equal to the first of its size, and the count of
# sumchecks the prover's own check did not accept (layer_check_failures): a direct measure of device-side slips.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r6soak
mkdir -p $OUT
cd $ROOT
T=${1:-60}
{
echo "== lanes, mixed sizes, defaults ($T s)"; timeout 600 python tools/stress.py $T | tail -1
echo "== one proof at a time, the solo paths ($T s)"; timeout 600 python tools/stress_solo.py $T | tail -1
echo "== GMiMC circuit lanes ($T s)"; timeout 600 python tools/stress_gmimc.py $T | tail -1
echo "== twelve lanes of bN = 18, look-ahead, pre-launch and round 0 ahead forced, thread cap 2^15"; GKRHIP_GMAX=15 GKRHIP_PRELAUNCH=2 GKRHIP_PRE=2 GKRHIP_COOP=2 GKRHIP_SPEC=0 GKRHIP_AHEAD=2 timeout 600 python tools/stress_one_size.py 18 12 120 | tail -1
echo "== twelve lanes of bN = 18, every solo path and speculation forced"; GKRHIP_PRELAUNCH=2 GKRHIP_PRE=2 GKRHIP_COOP=2 GKRHIP_SPEC=2 GKRHIP_AHEAD=2 timeout 600 python tools/stress_one_size.py 18 12 60 | tail -1
echo "== twenty-four lanes of bN = 20, defaults"; timeout 600 python tools/stress_one_size.py 20 24 20 | tail -1
echo "== forty-eight lanes of bN = 19, defaults: single calls that meet are proven in groups of 3"; timeout 600 python tools/stress_one_size.py 19 48 15 | tail -1
echo "== lanes, mixed sizes, every solo path forced ($T s)"; GKRHIP_PRELAUNCH=2 GKRHIP_PRE=2 GKRHIP_COOP=2 GKRHIP_SPEC=0 GKRHIP_AHEAD=2 timeout 600 python tools/stress.py $T | tail -1
echo "== sharded, 4 ranks on one GPU, shared-memory exchange (40 s)"; timeout 600 python tools/stress_sharded.py 40 4 shm | tail -2
} > $OUT/soaks.txt 2>&1
cat $OUT/soaks.txt | cut -c1-600
