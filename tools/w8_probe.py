"""Timing probe: 8 sharded ranks time-sharing one GPU, with and without a parent process that holds a HIP context."""
import importlib
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(world, sizes, env=None):
    name = "/gkrhip_p_" + uuid.uuid4().hex[:10]
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_shard_worker.py"), "shm", str(world), str(r), name, sizes],
                           env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    rcs = [p.wait() for p in ps]
    return time.time() - t0, rcs


print("no parent context: world 8, bN 6: %.1f s %s" % run(8, "6"))
print("no parent context: world 8, bN 24 digest: %.1f s %s" % run(8, "24", {"GKR_TEST_DIGEST": "1"}))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
print("parent holds a context: world 8, bN 6: %.1f s %s" % run(8, "6"))
s = [gk.MimcSession(10) for _ in range(6)]
print("parent holds a context and 6 sessions: world 8, bN 6: %.1f s %s" % run(8, "6"))
for x in s:
    x.close()
print("parent holds a context, 6 pooled lanes: world 8, bN 6: %.1f s %s" % run(8, "6"))
gk.shutdown()
print("parent after shutdown: world 8, bN 6: %.1f s %s" % run(8, "6"))
