import importlib, os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if mode in ("torch", "torchcomm"):
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(4, device="cuda")
import numpy as np
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
n = 3
if mode in ("comm", "torchcomm"):
    gk.comm_init_lanes(1, 0, np.stack([gk.comm_unique_id() for _ in range(n)]))
if mode == "commgone":
    gk.comm_init_lanes(1, 0, np.stack([gk.comm_unique_id() for _ in range(n)]))
    gk.comm_destroy()
if mode == "comm1":
    gk.comm_init_lanes(1, 0, np.stack([gk.comm_unique_id() for _ in range(1)]))
if mode == "shm":
    gk.comm_init_shm_lanes(1, 0, n, "/gkrprobe%d" % os.getpid())
bn = 18
ss = []
for _ in range(n):
    s = gk.MimcSession(bn); s.synth_inputs(); s.assign(); ss.append(s)
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")))
from bench import random_fr_array_np
qp = random_fr_array_np(bn)
ss[0].prove(qp)
t = time.perf_counter(); ss[0].prove(qp); t1 = time.perf_counter() - t
def work(k):
    for _ in range(2): ss[k].prove(qp)
ths = [threading.Thread(target=work, args=(k,)) for k in range(n)]
t = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
t3 = time.perf_counter() - t
print("mode %s: single %.1f ms; 6 proofs on 3 lanes %.1f ms => %.1f ms/proof" % (mode, t1*1e3, t3*1e3, t3*1e3/6))
