#!/bin/bash
# kernel + memory-copy timeline of one proof alone (bN = 20): where the device-side time of a layer goes
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r5_timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/solo_once.py ${1:-20} 2 > $OUT/run.txt 2>&1
cd $ROOT
cat $OUT/run.txt | tail -5
k=$(ls $OUT/t/*/*kernel_trace.csv | head -1); m=$(ls $OUT/t/*/*memory_copy_trace.csv | head -1)
ls -la $OUT/t/*/
# keep only the last proof's worth (the tail) to stay below the merge limit
tail -n 4000 $k > $OUT/kernel_tail.csv; head -1 $k > $OUT/kernel_head.csv
tail -n 400 $m > $OUT/memcpy_tail.csv; head -1 $m > $OUT/memcpy_head.csv
rm -rf $OUT/t
