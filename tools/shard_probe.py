"""Run the sharded worker (world ranks on one GPU, shm transport) with an environment and report every rank's last lines.
Usage: python tools/shard_probe.py <world> <sizes> <circuit> [VAR=val ...]"""
import os, subprocess, sys, uuid
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
world, sizes, circuit = int(sys.argv[1]), sys.argv[2], sys.argv[3]
e = dict(os.environ, GKR_ORACLE_THREADS="2", GKR_TEST_CIRCUIT=circuit)
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    e[k] = v
name = "/gkrhip_probe_" + uuid.uuid4().hex[:10]
ps = [subprocess.Popen([sys.executable, os.path.join(here, "gpu_shard_worker.py"), "shm", str(world), str(r), name, sizes],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
outs = [p.communicate(timeout=900)[0] for p in ps]
ok = all(p.returncode == 0 for p in ps)
print("PROBE", "OK" if ok else "FAIL", sys.argv[1:])
if not ok:
    for r, o in enumerate(outs):
        print("  rank", r, "|", " | ".join(o.strip().splitlines()[-2:]))
