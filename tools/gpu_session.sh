#!/bin/bash
# One GPU-box session of the round: parity tests, default bench line, the transports at world = 1 (every round forced
# through the exchange: bench.py --pass <shm|rccl_one_lane|rccl_tick|rccl_lanes>), two ranks on the one GPU.
# Usage (from the repo root, through gpurun):  bash tools/gpu_session.sh <tag> [steps...]
TAG=${1:-s}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for step in "$@"; do
  case $step in
    tests) timeout 1800 python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log ;;
    tail4) GKRHIP_HOST_TAIL=4 timeout 600 python bench.py --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_tail4.json 2> $OUT/bench_tail4.err ;;
    tail6) GKRHIP_HOST_TAIL=6 timeout 600 python bench.py --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_tail6.json 2> $OUT/bench_tail6.err ;;
    solo_tail0) timeout 600 python bench.py --concurrent 1 --steps 4 --warmup 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_solo_tail0.json 2> $OUT/bench_solo_tail0.err ;;
    solo_tail5) GKRHIP_HOST_TAIL=5 timeout 600 python bench.py --concurrent 1 --steps 4 --warmup 1 --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_solo_tail5.json 2> $OUT/bench_solo_tail5.err ;;
    ubench) timeout 600 ./tools/ubench > $OUT/ubench.txt 2>&1; grep -E "wg/CU=(2|8)" $OUT/ubench.txt | grep -E "mont_raw|fr_mac|f52|fma64|add64f|lshladd64|mad64 " ;;
    gmimc) timeout 600 python bench.py --circuit gmimc --bn 22 --no-cpu-baseline > $OUT/bench_gmimc22.json 2> $OUT/bench_gmimc22.err ;;
    oneshot) timeout 600 python tools/pcie_inclusive.py 24 > $OUT/oneshot24.txt 2>&1; tail -3 $OUT/oneshot24.txt ;;
    bench) timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json ;;
    bench_hwq8) GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_hwq8.json 2> $OUT/bench_hwq8.err ;;
    rccl0) GKRHIP_FORCE_COLLECTIVE=1 timeout 600 python bench.py --pass rccl_one_lane --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_rccl_one_lane.json 2> $OUT/bench_rccl_one_lane.err ;;
    rccl1) GKRHIP_FORCE_COLLECTIVE=1 timeout 600 python bench.py --pass rccl_tick --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_rccl_tick.json 2> $OUT/bench_rccl_tick.err ;;
    rccl_lanes) GKRHIP_FORCE_COLLECTIVE=1 GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --pass rccl_lanes --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_rccl_lanes.json 2> $OUT/bench_rccl_lanes.err ;;
    shm1) GKRHIP_FORCE_COLLECTIVE=1 timeout 600 python bench.py --pass shm --no-cpu-baseline --no-micro --no-oneshot --no-configs > $OUT/bench_shm1.json 2> $OUT/bench_shm1.err ;;
    w8) for v in default; do      # (the spin/sleep policy of the waiting threads is the option wait_spin_us since round 6)
          E=""
          ( time env $E python - <<'PY'
import subprocess, sys, os, uuid
here = os.path.join(os.getcwd(), "tests")
name = "/gkrhip_t_" + uuid.uuid4().hex[:10]
ps = [subprocess.Popen([sys.executable, os.path.join(here, "gpu_shard_worker.py"), "shm", "8", str(r), name, "6"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(8)]
print([p.wait() for p in ps])
PY
          ) > $OUT/w8_$v.log 2>&1; tail -4 $OUT/w8_$v.log; done ;;
    bench2) timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 2 --device 0 --bn 22 --steps 4 --warmup 2 > $OUT/bench_2ranks_1gpu.json 2> $OUT/bench_2ranks_1gpu.err; tail -c 1500 $OUT/bench_2ranks_1gpu.json; tail -5 $OUT/bench_2ranks_1gpu.err ;;
    w8probe) timeout 900 python tools/w8_probe.py > $OUT/w8_probe.log 2>&1; cat $OUT/w8_probe.log ;;
    *) echo "unknown step $step" ;;
  esac
done
for f in $OUT/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"] / 1e6, "M/s", d["ms_per_step"], "ms/step; latency", d["config"]["single_proof_latency_ms"], d["config"].get("per_round_exchange", ""))
except Exception as e:
    print("no json:", e)
PY
done
