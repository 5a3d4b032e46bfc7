"""CPU multi-process tests (gloo): the sharded sumcheck protocol (world_size 2, 4 and 8) reproduces the
un-sharded oracle transcript; the product's host-side shard helpers agree with the oracle."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import coracle as c
import pyoracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_protocol_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world), os.path.join(ROOT, "tests", "dist_cpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DIST-OK world=%d" % world in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_host_shard_helpers():
    gk = importlib.import_module("gkr-mimc_amd")
    q = c.random_fr_array(5)
    tab = c.folded_eq_table(q[2:])                      # eq over the last 3 coordinates
    for rank in range(8):
        assert np.array_equal(gk.host_shard_seed(q[2:], rank)[0], tab[rank])
    assert np.array_equal(gk.host_shard_seed(q[5:], 0), c.from_u64(1))
    # limb-split lanes of a sum of elements reduce to the field sum
    rng = np.random.default_rng(3)
    vals = [int(v) for v in rng.integers(1, 1 << 62, 300)]
    arr = c.from_ints([v * v * v for v in vals])
    lanes = np.zeros(8, np.uint64)
    for row in arr:
        for j in range(4):
            lanes[2 * j] += np.uint64(int(row[j]) & 0xFFFFFFFF)
            lanes[2 * j + 1] += np.uint64(int(row[j]) >> 32)
    want = sum(v * v * v for v in vals) % o.Q
    assert c.to_ints(gk.host_limbsplit_reduce(lanes))[0] == want
    assert np.array_equal(gk.host_mimc_hash(arr[:9]), c.mimc_hash(arr[:9]))
