#!/bin/bash
# records of the final round-5 tree: GPU test suite, default bench line, traces of one proof alone, the one-shot call
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r5final
mkdir -p $OUT
cd $ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $OUT/gpu_tests.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python tools/trace_rounds.py 20 2 > $OUT/bn20_solo_trace.txt 2>&1
python tools/trace_rounds.py 24 2 > $OUT/bn24_solo_trace.txt 2>&1
GKRHIP_TRACE=1 python tools/pcie_inclusive.py 24 2>&1 | grep -E "oneshot|one-shot|Verify" > $OUT/oneshot.txt
cat $OUT/gpu_tests.txt; tail -c 400 $OUT/bench_default.json; grep prove $OUT/bn20_solo_trace.txt $OUT/bn24_solo_trace.txt; cat $OUT/oneshot.txt
