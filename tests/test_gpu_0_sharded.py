"""GPU tests of the sharded prover (run with -m gpu): several ranks, one process each, time-sharing the ONE GPU of the
box and exchanging the per-round sums through the library's shared-memory transport (RCCL cannot form a communicator
of several ranks on one GPU); a 1-rank RCCL communicator with every round forced through ncclAllReduce.

This file sorts before test_gpu_parity.py on purpose and never touches the GPU from the pytest process itself: a
parent process that owns hardware queues pushes the device into time-slicing its queues across processes, which makes
the ranks' rounds ~40x slower (tools/w8_probe.py, profiles/r02_w8_probe_parent_context.txt: 2.5 s against minutes for a
full bN = 24 proof over 8 ranks)."""
import pytest

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- sharded prover (multi-process, one GPU)
def _run_shards(mode, world, sizes, env=None, collect=None):
    import os, subprocess, sys, uuid
    here = os.path.dirname(os.path.abspath(__file__))
    name = "/gkrhip_test_" + uuid.uuid4().hex[:12]
    e = dict(os.environ, GKR_ORACLE_THREADS="2")
    e.update(env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "gpu_shard_worker.py"), mode, str(world), str(r), name,
                               sizes], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    try:
        os.unlink("/dev/shm" + name)
    except OSError:
        pass
    bad = [r for r, (p, out) in enumerate(zip(procs, outs)) if p.returncode != 0 or "SHARD-OK" not in out]
    # the rank that failed first is the interesting one: its peers only report that somebody left
    bad.sort(key=lambda r: "a peer rank failed or left" in outs[r])
    assert not bad, "ranks %s failed; rank %d:\n%s" % (bad, bad[0], outs[bad[0]][-3000:])
    if collect is not None:
        collect.extend(outs)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_prover_matches_oracle(world):
    """shard on the lowest index bits; bN from log2(world) (no local round at all) upwards."""
    g = world.bit_length() - 1
    _run_shards("shm", world, ",".join(str(b) for b in sorted({g, g + 1, g + 2, 7, 10} if world < 8 else {3, 4, 8, 11})))


def test_sharded_prover_generic_path_and_small_budget():
    _run_shards("shm", 2, "3,6,9", {"GKRHIP_GENERIC": "1"})
    _run_shards("shm", 4, "4,9,11", {"GKRHIP_GMAX": "8"})


def test_sharded_prover_concurrent_lanes():
    """Two lanes per rank, each with its own collective channel, two proofs in flight per rank."""
    _run_shards("shm", 2, "4,9,11", {"GKR_TEST_LANES": "2"})
    _run_shards("rccl", 1, "3,9", {"GKR_TEST_LANES": "3", "GKRHIP_FORCE_COLLECTIVE": "1"})


def test_sharded_retry_after_a_missed_challenge():
    """One of four ranks withholds a challenge from its pre-launched round kernel (test hook, armed through
    gkrhip_set_option): the kernel gives up after a second, the rank votes in the round's exchange, EVERY rank leaves the
    layer's rounds at that exchange and runs them again in safe mode -- same transcript as the oracle, one retry per rank.
    Over both host-side exchanges: shared memory and the ticker."""
    env = {"GKRHIP_PRELAUNCH": "2", "GKR_TEST_DROP_RANK": "2", "GKR_TEST_DROP_ROUND": "3",
           "GKR_TEST_EXPECT_RETRIES": "1"}
    _run_shards("shm", 4, "12", env)
    _run_shards("tickshm", 2, "11", dict(env, GKR_TEST_DROP_RANK="1"))


def test_sharded_layer_rerun_after_a_corrupted_sum():
    """One bit of ONE rank's sums of one round flips before the exchange (test hook): the summed words are wrong on every
    rank, every rank's check of the finished sumcheck fails (it is a function of exchanged data: no vote), every rank runs
    the layer again in safe mode -- the oracle's transcript, one failure counted per rank.  Shared memory (4 ranks), the
    ticker (2 ranks), and a 1-rank RCCL communicator with every round through ncclAllReduce."""
    env = {"GKR_TEST_CORRUPT_RANK": "2", "GKR_TEST_CORRUPT_ROUND": "2", "GKR_TEST_EXPECT_LAYER_FAILURES": "1"}
    _run_shards("shm", 4, "12", env)
    _run_shards("shm", 4, "11", dict(env, GKR_TEST_CORRUPT_RANK="all", GKR_TEST_CORRUPT_LAYER="91"))     # the 91-claim key-copy layer (reference-shaped rounds: flipped after the exchange, so on every rank)
    _run_shards("tickshm", 2, "11", dict(env, GKR_TEST_CORRUPT_RANK="1", GKR_TEST_CORRUPT_ROUND="0"))
    _run_shards("rccl", 1, "10", dict(env, GKR_TEST_CORRUPT_RANK="all", GKRHIP_FORCE_COLLECTIVE="1"))


def test_sharded_oneshot_on_regular_form_buffers():
    """gkrhip_gkr_prove_mimc_regular with a communicator installed: the regular-form scope covers the boundary images
    only, the gathered Montgomery elements of the sharded phase 2 (multi-claim key-copy layer, host tail off) are uploaded
    as they are."""
    _run_shards("shm", 2, "1,2,5,9", {"GKR_TEST_REGULAR": "1"})
    _run_shards("shm", 4, "2,6,10", {"GKR_TEST_REGULAR": "1", "GKRHIP_HOST_TAIL": "0"})
    # with the hint's debug-mode check (hints.go:224-228): every rank verifies the sharded proof against its shards before returning
    _run_shards("shm", 2, "3,8", {"GKR_TEST_REGULAR": "1", "GKR_TEST_VERIFY_AFTER": "1"})


def test_rccl_ticker_world1_forced():
    """The multi-lane RCCL transport (one communicator, one issuing thread, batched ticks) with every round forced through
    it at world = 1: one lane and eight lanes in flight, MiMC and GMiMC (cipher, linear and multi-claim layers, the
    per-layer gather), against the oracle."""
    env = {"GKRHIP_FORCE_COLLECTIVE": "1"}
    _run_shards("tick", 1, "2,3,9,12", env)
    _run_shards("tick", 1, "4,9,11", dict(env, GKR_TEST_LANES="8"))
    _run_shards("tick", 1, "3,9", dict(env, GKR_TEST_CIRCUIT="gmimc"))
    _run_shards("tick", 1, "5,10", dict(env, GKRHIP_GENERIC="1", GKR_TEST_LANES="2"))
    _run_shards("tick", 1, "20", dict(env, GKR_TEST_DIGEST="1"))


def test_ticker_multi_rank_over_shared_memory():
    """The ticker with 2, 4 and 8 ranks on the one GPU (its tick all-reduce done on the host through shared memory; everything
    else -- slots, counts, contributing again until every rank is there, lanes of different ranks out of phase, the chunked
    gather, the votes to stop -- is the code an 8-GPU run executes): one and several lanes per rank, MiMC and GMiMC, against the
    oracle, and the bN = 22 digest over 2 ranks."""
    _run_shards("tickshm", 2, "1,2,5,9,11")
    _run_shards("tickshm", 4, "2,3,8,11")
    _run_shards("tickshm", 8, "3,4,9")
    _run_shards("tickshm", 2, "4,9,11", {"GKR_TEST_LANES": "3"})
    _run_shards("tickshm", 4, "9,10", {"GKR_TEST_LANES": "2", "GKRHIP_HOST_TAIL_SHARDED": "0"})
    _run_shards("tickshm", 2, "3,9", {"GKR_TEST_CIRCUIT": "gmimc"})
    _run_shards("tickshm", 2, "22", {"GKR_TEST_DIGEST": "1"})
    _run_shards("tick", 1, "9,12", {"GKRHIP_FORCE_COLLECTIVE": "1", "GKRHIP_TICK_DEVICE_BUF": "1", "GKR_TEST_LANES": "4"})


def test_sharded_prover_full_size_digests():
    """BASELINE config 3's size through the sharded driver: 8 ranks time-sharing the GPU (2^21-entry shards, the
    per-round exchange over shared memory) at bN = 24, and 2 ranks at bN = 22; the transcript must be the one the C
    oracle produced for the un-sharded proof (tests/golden/gkr_mimc_big_digests.json)."""
    _run_shards("shm", 8, "24", {"GKR_TEST_DIGEST": "1"})
    _run_shards("shm", 2, "22", {"GKR_TEST_DIGEST": "1"})


def test_config4_size_bn26_sharded_equals_unsharded():
    """BASELINE config 4's problem, bN = 26 (2^26 hashes; the reference itself stops at 2^24, poly/pool.go:13): one proof
    over 8 shards of 2^23 entries -- the ranks time-sharing the one GPU, 200 GB of resident shards -- and the same proof
    un-sharded on the one GPU (186 GB of tables).  Every rank's transcript and the un-sharded transcript have the same
    SHA-256, the native gkr.Verify accepts it on every rank and rejects a corrupted copy.  No oracle reaches this size;
    the sharded and un-sharded drivers are pinned against the oracle at bN <= 24 by the tests above."""
    import re
    one, eight = [], []
    _run_shards("shm", 1, "26", {"GKR_TEST_HASHONLY": "1"}, collect=one)
    _run_shards("shm", 8, "26", {"GKR_TEST_HASHONLY": "1"}, collect=eight)
    shas = {m.group(1) for out in one + eight for m in re.finditer(r"SHA bn=26 ([0-9a-f]{64})", out)}
    assert len(shas) == 1 and sum(len(re.findall(r"SHA bn=26", out)) for out in one + eight) == 9, shas


def test_sharded_gmimc_circuit():
    """BASELINE config 5's circuit sharded (linear layers included): small sizes against the C oracle's un-sharded
    transcript, bN = 14 and 20 against the committed digests."""
    _run_shards("shm", 2, "1,2,5,9", {"GKR_TEST_CIRCUIT": "gmimc"})
    _run_shards("shm", 4, "2,3,8,11", {"GKR_TEST_CIRCUIT": "gmimc"})
    _run_shards("shm", 4, "14,20", {"GKR_TEST_CIRCUIT": "gmimc", "GKR_TEST_DIGEST": "1"})
    _run_shards("shm", 2, "3,6,9", {"GKR_TEST_CIRCUIT": "gmimc", "GKRHIP_GENERIC": "1"})
    # the fused linear rounds through ncclAllReduce (1-rank communicator, every round forced through the collective)
    _run_shards("rccl", 1, "2,9", {"GKR_TEST_CIRCUIT": "gmimc", "GKRHIP_FORCE_COLLECTIVE": "1"})
    # registered 1-, 3- and 4-input gates sharded
    _run_shards("shm", 4, "2,3,7,10", {"GKR_TEST_CIRCUIT": "variadic"})


def test_rccl_plumbing_world1():
    """RCCL is dlopen()ed, a 1-rank communicator is created and every round's sums go through
    ncclAllReduce (GKRHIP_FORCE_COLLECTIVE): the call sequence of the multi-GPU path on the one GPU we have."""
    _run_shards("rccl", 1, "1,2,5,9", {"GKRHIP_FORCE_COLLECTIVE": "1"})




def test_sharded_host_tail_gather():
    """GKRHIP_HOST_TAIL_SHARDED = h: the ranks gather the tables of the round with 2^(h+1) pairs and every rank finishes the
    remaining local rounds AND the rounds over the shard bits on the host -- same transcript for h = 0 (every local round
    exchanged), 1, 4 (default), 6, over the shared-memory exchange and through the ticker / a per-lane communicator at world 1."""
    for h in ("0", "1", "6"):
        _run_shards("shm", 4, "8,9,12", {"GKRHIP_HOST_TAIL_SHARDED": h})
    _run_shards("shm", 2, "7,11", {"GKRHIP_HOST_TAIL_SHARDED": "3", "GKRHIP_GMAX": "8"})
    _run_shards("shm", 8, "10,12", {"GKRHIP_HOST_TAIL_SHARDED": "2"})
    _run_shards("tick", 1, "9,12", {"GKRHIP_FORCE_COLLECTIVE": "1", "GKRHIP_HOST_TAIL_SHARDED": "5"})     # chunked gather through the ticker
    _run_shards("rccl", 1, "9,11", {"GKRHIP_FORCE_COLLECTIVE": "1", "GKRHIP_HOST_TAIL_SHARDED": "3"})
    # the fused linear rounds (add / copy layers of the GMiMC circuit, registered 1-, 3- and 4-input gates) gather as well
    _run_shards("shm", 4, "8,10", {"GKR_TEST_CIRCUIT": "gmimc", "GKRHIP_HOST_TAIL_SHARDED": "2"})
    _run_shards("shm", 2, "9", {"GKR_TEST_CIRCUIT": "gmimc", "GKRHIP_HOST_TAIL_SHARDED": "6"})
    _run_shards("shm", 2, "13,14", {"GKRHIP_HOST_TAIL_SHARDED": "10"})                          # the raised maximum
    _run_shards("shm", 4, "9,10", {"GKR_TEST_CIRCUIT": "variadic", "GKRHIP_HOST_TAIL_SHARDED": "3"})
    _run_shards("tick", 1, "9", {"GKRHIP_FORCE_COLLECTIVE": "1", "GKR_TEST_CIRCUIT": "gmimc", "GKRHIP_HOST_TAIL_SHARDED": "4"})


def test_sharded_host_tail_off():
    """GKRHIP_HOST_TAIL=0: the gathered tail rounds of a sharded sumcheck on the device instead of the host."""
    _run_shards("shm", 4, "3,4,9,11", {"GKRHIP_HOST_TAIL": "0"})


def _run_shards_expect_failure(world, sizes, env, within_s, mode="shm"):
    """Rank GKR_TEST_DIE leaves before proving: every other rank must end with an error inside `within_s` seconds."""
    import os, subprocess, sys, time, uuid
    here = os.path.dirname(os.path.abspath(__file__))
    name = "/gkrhip_test_" + uuid.uuid4().hex[:12]
    e = dict(os.environ, GKR_ORACLE_THREADS="2")
    e.update(env)
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "gpu_shard_worker.py"), mode, str(world), str(r), name, sizes],
                              env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=within_s)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung after its peer left")
        outs.append(out)
    die = int(env.get("GKR_TEST_DIE", "-1"))
    for r, (p, out) in enumerate(zip(procs, outs)):
        if r == die:
            continue
        assert p.returncode != 0 and "SHARD-OK" not in out, "rank %d:\n%s" % (r, out[-2000:])
        assert "peer rank failed or left" in out or "timed out" in out, out[-2000:]
    return time.time() - t0


def test_peer_failure_is_an_error_not_a_hang():
    """the shared-memory barrier has an abort word (raised by a rank that leaves in an orderly way) and a
    deadline (for a rank that is killed): the survivors fail with an error either way."""
    _run_shards_expect_failure(2, "9", {"GKR_TEST_DIE": "1"}, within_s=120)
    dt = _run_shards_expect_failure(4, "8", {"GKR_TEST_DIE": "2", "GKR_TEST_DIE_HARD": "1", "GKRHIP_COLL_TIMEOUT_S": "4"}, within_s=120)
    assert dt < 60


def test_ticker_lane_timeout_fails_every_rank():
    """a lane that gives up waiting in the ticker must not leave its words behind for a late peer to complete
    ITS exchange against.  Rank 1 arrives after rank 0's collective time-out: rank 0 fails, its ticker stops, and rank 1
    fails too (no proof on the stale payload of an abandoned round)."""
    dt = _run_shards_expect_failure(2, "9", {"GKR_TEST_DELAY_RANK": "1", "GKR_TEST_DELAY_S": "9", "GKRHIP_COLL_TIMEOUT_S": "3"},
                                    within_s=180, mode="tickshm")
    assert dt < 120


def test_bench_captures_rccl_warnings_of_a_failed_pass(tmp_path):
    """first contact with a real multi-rank communicator happens on the driver's 8-GPU node, where nobody can
    re-run by hand -- so the one RCCL failure this pool CAN provoke (two ranks of one communicator on ONE GPU: "Duplicate GPU
    detected", ncclInvalidUsage) must leave RCCL's own warning in the bench line (`passes.rccl_one_lane.rccl_log`), not only
    "invalid usage".  bench.py as the driver launches it for N = 2, both ranks on device 0, the RCCL pass alone."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, GKRHIP_BENCH_PASS_LIMIT_S="240")
    env.pop("NCCL_DEBUG", None)
    env.pop("NCCL_DEBUG_FILE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--device", "0", "--bn", "16", "--steps", "2",
           "--warmup", "1", "--passes", "rccl_one_lane", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    line = None
    for l in out.stdout.strip().splitlines()[::-1]:
        try:
            line = json.loads(l)
            break
        except Exception:
            continue
    assert line is not None, out.stdout[-1500:] + out.stderr[-3000:]
    p = line["passes"]["rccl_one_lane"]
    # one GPU cannot host two ranks of a communicator: bench.py's rank 0 exits with code 3, which the launcher reports as a failure
    assert "error" in p and out.returncode != 0, (p, out.returncode)
    assert line.get("degraded") == "all_passes_failed" and line["n_gpus_rccl"] == 0
    log = p.get("rccl_log", "")
    assert "NCCL WARN" in log, "no RCCL warning captured:\n%s\n--- stderr\n%s" % (json.dumps(p)[:1500], out.stderr[-3000:])
    assert "bench.py: pass" not in log                                        # RCCL's words, not bench.py's own
