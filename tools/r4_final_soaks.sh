#!/bin/bash
# Soaks of the final tree (run on the GPU box through gpurun): every proof byte-equal to the first of its size.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4soak
mkdir -p $OUT
cd $ROOT
{
echo "== lanes, mixed sizes, defaults (120 s)"; timeout 400 python tools/stress.py 120 | tail -1
echo "== one proof at a time, the solo paths (90 s)"; timeout 400 python tools/stress_solo.py 90 | tail -1
echo "== GMiMC circuit lanes (60 s)"; timeout 400 python tools/stress_gmimc.py 60 | tail -1
echo "== twelve lanes of bN = 18, look-ahead and pre-launch forced, thread cap 2^15"; GKRHIP_GMAX=15 GKRHIP_PRELAUNCH=2 GKRHIP_PRE=2 GKRHIP_COOP=2 GKRHIP_PRELAUNCH_LG=30 GKRHIP_SPEC=0 timeout 400 python tools/stress_one_size.py 18 12 120 | tail -1
echo "== twenty-four lanes of bN = 20, defaults"; timeout 400 python tools/stress_one_size.py 20 24 20 | tail -1
echo "== lanes, mixed sizes, every solo path forced (60 s)"; GKRHIP_PRELAUNCH=2 GKRHIP_PRE=2 GKRHIP_COOP=2 GKRHIP_PRELAUNCH_LG=30 GKRHIP_SPEC=0 timeout 400 python tools/stress.py 60 | tail -1
echo "== sharded, 4 ranks on one GPU, shared-memory exchange (40 s)"; timeout 400 python tools/stress_sharded.py 40 4 shm | tail -2
} > $OUT/soaks.txt 2>&1
cat $OUT/soaks.txt | cut -c1-400
