#!/bin/bash
# The thread cap of the round kernels against the number of proofs in flight (run on the GPU box through gpurun):
# GKRHIP_GMAX = 15 | 16 forced, and the library's own choice (16 below ten proofs in flight, 15 from ten).
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'wait', round(d['host_split_ms_per_step']['host_wait_ms'],1))"; }
COMMON="--no-cpu-baseline --no-micro --no-oneshot --no-configs"
run() { name=$1; shift; for g in 15 16 auto; do echo -n "$name GMAX=$g: "; if [ $g = auto ]; then python bench.py "$@" $COMMON 2>/dev/null | val; else GKRHIP_GMAX=$g python bench.py "$@" $COMMON 2>/dev/null | val; fi; done; }
run bn20x24 --bn 20 --concurrent 24 --steps 48 --warmup 24
run bn20x8 --bn 20 --concurrent 8 --steps 32 --warmup 8
run bn22x8 --bn 22 --concurrent 8 --steps 24 --warmup 8
run bn22x12 --bn 22 --concurrent 12 --steps 24 --warmup 12
run gmimc22x12 --circuit gmimc --bn 22 --concurrent 12 --steps 24 --warmup 12
run bn24x5 --steps 10 --warmup 5
