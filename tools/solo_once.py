"""One proof alone on the GPU, a few times (for rocprofv3 timelines): python tools/solo_once.py <bn> [reps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
for kv in filter(None, os.environ.get("GKR_SOLO_OPTIONS", "").split(",")):      # library options for A/B runs
    gk.set_option(kv.split("=")[0], int(kv.split("=")[1]))
bn = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = gk.MimcSession(bn); s.synth_inputs(); s.assign()
qp = bench.random_fr_array_np(bn)
for _ in range(2): s.prove(qp)
for _ in range(reps):
    t0 = time.perf_counter(); s.prove(qp); print("prove %.2f ms" % (1e3 * (time.perf_counter() - t0)))
