#!/bin/bash
# rocprofv3 kernel stats of one MSM size with the one- and the two-level sort (run on the GPU box through gpurun):
#   bash tools/r4_msm_sort_prof.sh [logn]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
LOGN=${1:-24}
OUT=$ROOT/gpurun_out/r4sort
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lv in 1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l$lv -- python3 $ROOT/tools/msm_bench.py $LOGN levels=$lv > $OUT/msm${LOGN}_l$lv.txt 2>&1 < /dev/null
  f=$(ls /tmp/prof_l$lv/*/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/msm${LOGN}_l${lv}_kernel_stats.csv; head -16 "$f" | cut -c1-160; else echo "no stats for levels=$lv"; tail -5 $OUT/msm${LOGN}_l$lv.txt; fi
done
