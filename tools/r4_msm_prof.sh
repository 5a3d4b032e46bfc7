#!/bin/bash
# MSM measurements of the final tree (run on the GPU box through gpurun): micro-benchmarks G1 / G2, rocprofv3 kernel stats,
# witness-like scalars, PCIe-inclusive calls, the device side of ComputeGroth16Proof.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4msm
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/msm_bench.py 16 18 20 22 24 > $OUT/msm_g1.jsonl 2>&1 < /dev/null
timeout 300 python tools/msm_bench.py g2 20 22 > $OUT/msm_g2.jsonl 2>&1 < /dev/null
timeout 300 python tools/msm_witness_like.py > $OUT/witness_like.txt 2>&1 < /dev/null
timeout 300 python tools/msm_pcie_inclusive.py > $OUT/pcie_inclusive.txt 2>&1 < /dev/null
timeout 600 python tools/groth16_backhalf.py > $OUT/groth16_backhalf.txt 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_g1 -- python3 $ROOT/tools/msm_bench.py 20 22 24 > /dev/null 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_g2 -- python3 $ROOT/tools/msm_bench.py g2 20 22 > /dev/null 2>&1 < /dev/null
for t in g1 g2; do
  f=$(ls /tmp/p_$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $OUT/msm_${t}_kernel_stats.csv
done
tail -3 $OUT/*.txt | cut -c1-300
cut -c1-200 $OUT/msm_g1.jsonl $OUT/msm_g2.jsonl
