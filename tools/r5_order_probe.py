"""Why does the GMiMC job run 10 % slower behind a job of 56 small proofs?  (profiles/r05_order_probe.txt)
usage: python tools/r5_order_probe.py <variant>   variant: none | shutdown | solo_first"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
variant = sys.argv[1] if len(sys.argv) > 1 else "none"

def job(bn, lanes, per, layers=None):
    ss = []
    for _ in range(lanes):
        s = gk.MimcSession(bn, layers=layers); s.synth_inputs(); s.assign(); ss.append(s)
    qp = bench.random_fr_array_np(bn)
    def work(s, n):
        for _ in range(n): s.prove(qp)
    for rep, n in ((0, 1), (1, per)):
        th = [threading.Thread(target=work, args=(s, n)) for s in ss]
        gk.synchronize(); t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        gk.synchronize(); dt = time.perf_counter() - t0
    for s in ss: s.close()
    return (1 << bn) * lanes * per / dt / 1e6

g = gk.gmimc_t2_circuit()
if variant == "solo_first":
    print("gmimc 12 lanes, fresh process: %.1f M/s" % job(22, 12, 4, g))
print("bn20 56 lanes: %.1f M/s" % job(20, 56, 3))
if variant == "shutdown":
    gk.shutdown(); gk.init(0)
print("gmimc 12 lanes behind it (%s): %.1f M/s" % (variant, job(22, 12, 4, g)))
print("gmimc again: %.1f M/s" % job(22, 12, 4, g))
