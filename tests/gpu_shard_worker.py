"""One rank of the sharded GKR prover, run by the GPU tests: `world` processes time-share GPU 0 and
exchange the per-round limb-split sums through the library's shared-memory transport (RCCL cannot form a
communicator of several ranks on one GPU).  Exercises the real C++ sharded driver (shard weights, per-round
collective, gather, redundant tail rounds) and checks the transcript against the un-sharded oracle.
With world == 1 and mode == rccl it exercises the RCCL plumbing (dlopen, communicator, all-reduce calls)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import coracle as c  # noqa: E402


def concurrent(gk, world, rank, sizes, nlanes):
    """nlanes sessions (session k on lane k on every rank) proving concurrently from nlanes host threads."""
    import threading
    for bn in sizes:
        n = 1 << bn
        i0, qp = c.random_fr_array(n), c.random_fr_array(bn)
        want = c.gkr_prove_mimc(bn, i0, i0.copy(), qp)[0]
        ss = []
        for _ in range(nlanes):
            s = gk.MimcSession(bn)
            s.synth_inputs()
            s.assign()
            ss.append(s)
        got = [[None, None] for _ in range(nlanes)]

        def work(k):
            for rep in range(2):
                got[k][rep] = ss[k].prove(qp)

        ths = [threading.Thread(target=work, args=(k,)) for k in range(nlanes)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for k in range(nlanes):
            for rep in range(2):
                assert np.array_equal(got[k][rep], want), ("lanes", bn, rank, k, rep)
            ss[k].close()
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d lanes=%d %s" % (rank, world, nlanes, sizes))


def digests(gk, world, rank, sizes, circuit):
    """BASELINE sizes: RandomFrArray inputs generated on the device (this rank's shard), transcript SHA-256 against
    the digest the C oracle produced for the UN-sharded proof (tests/golden/*_big_digests.json)."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden",
                                       "gkr_gmimc_big_digests.json" if circuit == "gmimc" else "gkr_mimc_big_digests.json")))
    layers = gk.gmimc_t2_circuit() if circuit == "gmimc" else None
    for bn in sizes:
        want = [e for e in gold if e["bn"] == bn][0]
        s = gk.MimcSession(bn, layers=layers)
        s.synth_inputs()
        s.assign()
        qp = c.random_fr_array(bn)
        flat = s.prove(qp)
        assert flat.shape[0] == want["n_elements"]
        assert hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest() == want["sha256_flat"], ("digest", circuit, bn, rank)
        assert s.verify(qp, flat), ("verify", circuit, bn, rank)
        s.close()
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d digests %s %s" % (rank, world, circuit, sizes))


def hash_only(gk, world, rank, sizes):
    """Sizes no oracle reaches (bN = 26: BASELINE config 4): RandomFrArray inputs generated on the device, proof accepted
    by the native gkr.Verify against the resident shards, a corrupted proof rejected; prints the transcript's SHA-256 so
    that the parent can compare the sharded runs with the un-sharded one."""
    import hashlib
    for bn in sizes:
        s = gk.MimcSession(bn)
        s.synth_inputs()
        s.assign()
        qp = c.random_fr_array(bn)
        flat = s.prove(qp)
        assert s.verify(qp, flat), ("verify", bn, rank)
        bad = flat.copy()
        bad[len(bad) // 2, 1] ^= np.uint64(2)
        assert not s.verify(qp, bad), ("corrupted proof accepted", bn, rank)
        print("SHA bn=%d %s" % (bn, hashlib.sha256(flat.astype("<u8").tobytes()).hexdigest()))
        s.close()
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d hash-only %s" % (rank, world, sizes))


def regular_oneshot(gk, world, rank, sizes):
    """The hint-shaped one-shot call on REGULAR-form buffers with a communicator installed (the session of the call
    shards): this rank's shard of the inputs in, the un-sharded oracle transcript (converted to regular form) out."""
    for bn in sizes:
        n = 1 << bn
        i0 = c.random_fr_array(n)
        i1 = c.from_ints([(5 * i * i + 11 * i + 3) for i in range(n)])
        qp = c.random_fr_array(bn)
        oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i1, qp)
        def reg(a):       # Montgomery elements -> the words of their regular form
            return np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in c.to_ints(a)], dtype=np.uint64).reshape(-1, 4)
        flat, outs = gk.gkr_prove_mimc(reg(i0[rank::world].copy()), reg(i1[rank::world].copy()), reg(qp), regular=True)
        assert np.array_equal(flat, reg(oflat)), ("regular transcript", bn, rank)
        assert np.array_equal(outs, reg(oouts[rank::world].copy())), ("regular outputs", bn, rank)
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d regular %s" % (rank, world, sizes))


def gmimc_small(gk, world, rank, sizes):
    """The GMiMC (t = 2) circuit sharded: cipher, add and identity layers against the C oracle's un-sharded transcript."""
    import pyoracle as o
    layers = gk.gmimc_t2_circuit()
    descs = c.circuit_descs(o.gmimc_t2_circuit())
    for bn in sizes:
        n = 1 << bn
        ins = [c.random_fr_array(n) if i % 2 == 0 else c.from_ints([(7 * j * j + i) % 1000003 for j in range(n)]) for i in range(4)]
        qp = c.random_fr_array(bn)
        s = gk.MimcSession(bn, layers=layers)
        for i, t in enumerate(ins):
            s.load_input(i, t[rank::world].copy())
        s.assign()
        flat = s.prove(qp)
        oflat, oouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
        assert np.array_equal(flat, oflat), ("gmimc transcript", bn, rank)
        assert np.array_equal(s.outputs(), oouts[rank::world]), ("gmimc outputs", bn, rank)
        assert s.verify(qp, flat)
        s.close()
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d gmimc %s" % (rank, world, sizes))


def variadic(gk, world, rank, sizes):
    """A circuit over registered 1-, 3- and 4-input gates, sharded, against the C oracle's un-sharded transcript."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_parity as T
    circ, layers = T._variadic_circuit()
    descs = c.circuit_descs(circ)
    for bn in sizes:
        n = 1 << bn
        ins = [c.random_fr_array(n), T.nasty(n, bn + 5), c.from_ints([(3 * j * j + 1) % 1000003 for j in range(n)]), T.nasty(n, bn + 6)]
        qp = c.random_fr_array(bn)
        s = gk.MimcSession(bn, layers=layers)
        for i, t in enumerate(ins):
            s.load_input(i, t[rank::world].copy())
        s.assign()
        flat = s.prove(qp)
        oflat, oouts, _ = c.gkr_prove_circuit(descs, bn, ins, qp)
        assert np.array_equal(flat, oflat), ("variadic transcript", bn, rank)
        assert np.array_equal(s.outputs(), oouts[rank::world]), ("variadic outputs", bn, rank)
        assert s.verify(qp, flat)
        s.close()
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d variadic %s" % (rank, world, sizes))


def main():
    mode, world, rank, name = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sizes = [int(x) for x in sys.argv[5].split(",")]
    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(0)
    nlanes = int(os.environ.get("GKR_TEST_LANES", "1"))
    if mode == "shm":
        gk.comm_init_shm_lanes(world, rank, nlanes, name)
    elif mode == "tickshm":    # the ticker's logic with several ranks on the one GPU: its tick is a host all-reduce through shared memory
        gk.comm_init_tick_shm(world, rank, nlanes, name + "_tick")
    elif mode == "tick":       # all lanes over one RCCL communicator (1-rank communicator: every exchange is a real ncclAllReduce)
        gk.comm_init_tick(1, 0, nlanes, gk.comm_unique_id())
    else:
        gk.comm_init_lanes(1, 0, np.stack([gk.comm_unique_id() for _ in range(nlanes)]))
    if os.environ.get("GKR_TEST_DIE") == str(rank):
        # a rank that leaves without proving: its peers must fail with an error, not hang
        if os.environ.get("GKR_TEST_DIE_HARD"):
            os._exit(3)                     # killed: nobody raises the abort word, the deadline has to
        gk.comm_destroy()                   # orderly exit: raises the abort word of the segment
        print("DIED rank %d" % rank)
        return
    circuit = os.environ.get("GKR_TEST_CIRCUIT", "mimc")
    if os.environ.get("GKR_TEST_DELAY_RANK") == str(rank):
        import time
        time.sleep(float(os.environ.get("GKR_TEST_DELAY_S", "5")))      # this rank reaches its first exchange late
    if os.environ.get("GKR_TEST_DROP_RANK") == str(rank):
        # this rank withholds ONE challenge from its pre-launched round kernel: the kernel gives up after a second, the rank
        # votes for a retry in the round's exchange and every rank runs the layer's rounds again (same transcript)
        gk.set_option("test_drop_challenge", int(os.environ.get("GKR_TEST_DROP_ROUND", "3")))
    if os.environ.get("GKR_TEST_CORRUPT_RANK") in (str(rank), "all"):
        # one bit of this rank's sums of one round flips before the ranks add them (GKR_TEST_CORRUPT_RANK=all: on every rank, for
        # the device-side exchange, whose words are flipped after the all-reduce): every rank sees a sumcheck that does not
        # close and every rank runs the layer again -- no vote needed, the verdict is a function of exchanged data
        gk.set_option("test_corrupt_sum", int(os.environ.get("GKR_TEST_CORRUPT_ROUND", "2")))
        gk.set_option("test_corrupt_skip", int(os.environ.get("GKR_TEST_CORRUPT_LAYER", "3")))
    if os.environ.get("GKR_TEST_VERIFY_AFTER"):
        gk.set_option("verify_after_prove", 1)      # the one-shot calls run gkr.Verify on the (sharded) proof before returning it
    if os.environ.get("GKR_TEST_EXPECT_RETRIES") or os.environ.get("GKR_TEST_EXPECT_LAYER_FAILURES"):
        gk.profile_reset(0)
    if os.environ.get("GKR_TEST_REGULAR"):
        return regular_oneshot(gk, world, rank, sizes)
    if os.environ.get("GKR_TEST_HASHONLY"):
        return hash_only(gk, world, rank, sizes)
    if os.environ.get("GKR_TEST_DIGEST"):
        return digests(gk, world, rank, sizes, circuit)
    if circuit == "gmimc":
        return gmimc_small(gk, world, rank, sizes)
    if circuit == "variadic":
        return variadic(gk, world, rank, sizes)
    if nlanes > 1:
        return concurrent(gk, world, rank, sizes, nlanes)
    for bn in sizes:
        n = 1 << bn
        i0 = c.random_fr_array(n)
        i1 = c.from_ints([(5 * i * i + 11 * i + 3) for i in range(n)])
        qp = c.random_fr_array(bn)
        s = gk.MimcSession(bn)
        if bn % 2:
            s.load_inputs(i0[rank::world].copy(), i1[rank::world].copy())
        else:
            i1 = i0.copy()
            s.synth_inputs()
        s.assign()
        flat = s.prove(qp)
        oflat, oouts, _ = c.gkr_prove_mimc(bn, i0, i1, qp)
        assert np.array_equal(flat, oflat), ("transcript", bn, rank)
        assert np.array_equal(s.outputs(), oouts[rank::world]), ("outputs", bn, rank)
        pt = c.mimc_hash(c.from_u64(bn)).repeat(bn, axis=0) if bn else c.fr(0)
        assert np.array_equal(s.evaluate_layer(93, pt), c.evaluate(oouts, pt)), ("evaluate", bn, rank)
        assert np.array_equal(flat, s.prove(qp)), ("repeat", bn, rank)
        s.close()
    if os.environ.get("GKR_TEST_EXPECT_RETRIES"):
        got = gk.profile_get()["chal_retries"]
        assert got == int(os.environ["GKR_TEST_EXPECT_RETRIES"]), ("chal_retries", rank, got)
    if os.environ.get("GKR_TEST_EXPECT_LAYER_FAILURES"):
        got = gk.profile_get()["layer_check_failures"]
        assert got == int(os.environ["GKR_TEST_EXPECT_LAYER_FAILURES"]), ("layer_check_failures", rank, got)
    gk.comm_destroy()
    print("SHARD-OK rank %d/%d %s" % (rank, world, sizes))


if __name__ == "__main__":
    main()
