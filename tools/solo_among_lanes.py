"""One proof alone on the GPU while N-1 other sessions (lanes, streams) exist but are idle -- the shape of bench.py's
`configs.*.single_proof_ms`.  Usage: python tools/solo_among_lanes.py <bn> <lanes> [reps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
bn, lanes = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ss = []
for _ in range(lanes):
    s = gk.MimcSession(bn); s.synth_inputs(); s.assign(); ss.append(s)
qp = bench.random_fr_array_np(bn)
import threading
def warm(s):
    for _ in range(2): s.prove(qp)
ths = [threading.Thread(target=warm, args=(s,)) for s in ss]
[t.start() for t in ths]; [t.join() for t in ths]
gk.synchronize()
for which in (range(lanes) if os.environ.get('SOLO_ALL') else (0, lanes - 1)):
    lat = []
    for _ in range(reps):
        gk.profile_reset(0)
        t0 = time.perf_counter(); ss[which].prove(qp); lat.append(1e3 * (time.perf_counter() - t0))
    p = gk.profile_get()
    print("bn %d, %d lanes alive, session %d alone: %s ms  (spec_rounds %d coop %d prelaunched %d)" % (bn, lanes, which, " ".join("%.1f" % x for x in lat), p["spec_rounds"], p["coop_rounds"], p["prelaunched_rounds"]))
