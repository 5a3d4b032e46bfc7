#!/usr/bin/env python3
"""Soak test of the sharded driver: `world` ranks time-sharing the GPU prove the same sizes over and over (several
lanes per rank); every transcript must equal the oracle's.  python tools/stress_sharded.py [seconds] [world] [shm|tickshm]"""
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    mode = sys.argv[3] if len(sys.argv) > 3 else "shm"       # shm | tickshm (the ticker's logic over shared memory)
    t_end = time.time() + budget
    runs = 0
    while time.time() < t_end:
        name = "/gkrhip_soak_" + uuid.uuid4().hex[:10]
        env = dict(os.environ, GKR_ORACLE_THREADS="2", GKR_TEST_LANES=str(1 + runs % 3))
        sizes = ["4,9,11", "3,10,12", "5,8,13"][runs % 3]
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_shard_worker.py"), mode, str(world), str(r), name, sizes],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate()[0] for p in ps]
        for r, (p, out) in enumerate(zip(ps, outs)):
            if p.returncode != 0 or "SHARD-OK" not in out:
                print("FAILED run %d rank %d:\n%s" % (runs, r, out[-3000:]))
                sys.exit(1)
        runs += 1
    print("sharded soak (%s): %d runs of world %d, all transcripts equal the oracle's" % (mode, runs, world))


if __name__ == "__main__":
    main()
