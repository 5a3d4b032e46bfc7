#!/bin/bash
# Round-4 profile session (run on the GPU box through gpurun): rocprofv3 kernel stats of the GKR bench (tools/prof_session.sh),
# the PMC passes of the fold (tools/pmc_bench.sh), kernel stats of the MSM (G1, G2) and of computeH.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4prof
mkdir -p $OUT
bash $ROOT/tools/prof_session.sh r4prof > $OUT/prof_session.log 2>&1
bash $ROOT/tools/pmc_bench.sh 24 > $OUT/pmc_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/msm_g1 -- python3 $ROOT/tools/msm_bench.py 20 22 24 > $OUT/msm_g1.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/msm_g2 -- python3 $ROOT/tools/msm_bench.py g2 20 22 > $OUT/msm_g2.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/computeh -- python3 $ROOT/tools/computeh_bench.py 20 22 24 > $OUT/computeh.txt 2>&1
cd $ROOT
for t in msm_g1 msm_g2 computeh; do
  f=$(ls $OUT/$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv
done
ls -la $OUT | head -40
