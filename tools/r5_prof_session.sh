#!/bin/bash
# Round-5 profile session (gpurun): rocprofv3 kernel stats of the GKR bench (one proof, five lanes), fold launches by size, PMC passes
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r5prof
mkdir -p $OUT
bash $ROOT/tools/prof_session.sh r5prof 2>&1 | grep -v "computeH\|ch_prof" > $OUT/prof_session.log
ls $OUT | head -30
cat $OUT/pmc_fold_traffic.json | head -20
