#!/usr/bin/env python3
"""bench.py -- headline benchmark: MiMC hashes GKR-proved per second (gkr.Prove on
examples.MimcCircuit, assignment excluded from the timer as BenchmarkGkr does,
gkr/gkr_test.go:99-105), plus the fold kernel's achieved HBM bandwidth and the CPU oracle timed
beside it.

    python bench.py --gpus N --steps K --warmup W [--bn B]

N = 1: one process, GPU 0.  N > 1: launched by torch.distributed.run, one rank per GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def cpu_baseline(target_seconds=40.0):
    """Time the CPU oracle (C restatement, OpenMP over the host cores) on a bounded sample of the same
    workload: gkr.Prove of 2^b MiMC hashes with RandomFrArray inputs; b grows until a run takes
    >= target_seconds/4 (each +1 doubles the work)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle
    best = None
    b = 14
    while True:
        i0 = coracle.random_fr_array(1 << b)
        qp = coracle.random_fr_array(b)
        _, _, secs = coracle.gkr_prove_mimc(b, i0, i0.copy(), qp, want_outputs=False)
        best = (b, secs)
        if secs >= target_seconds / 4 or b >= 20:
            break
        b += 2 if secs < target_seconds / 16 else 1
    b, secs = best
    return {"value": (1 << b) / secs, "unit": "MiMC hashes GKR-proved/s", "cores": coracle.lib.oracle_num_threads(),
            "kind": "port", "sample": "gkr.Prove of 2^%d hashes (RandomFrArray inputs), %.2f s, C restatement "
                                      "of the reference algorithm (not the Go binary)" % (b, secs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bn", type=int, default=24, help="log2 of the number of MiMC hashes per proof (per job)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--circuit", choices=["mimc", "gmimc"], default="mimc",
                    help="mimc: examples.MimcCircuit (the headline metric); gmimc: the build-defined GMiMC t=2 circuit "
                         "(BASELINE config 5, quoted at --bn 22)")
    ap.add_argument("--concurrent", type=int, default=5,
                    help="independent proofs in flight (each on its own resident session/lane/stream and, when "
                         "sharded, its own communicator); 1 = strictly one proof at a time")
    ap.add_argument("--exchange", choices=["auto", "rccl", "shm"], default="auto",
                    help="transport of the per-round 576-byte sum of the sharded prover: rccl = ncclAllReduce on the lane's "
                         "stream; shm = the ranks add the words on the host through POSIX shared memory (one node); auto = "
                         "shm when every rank is on this node, else rccl")
    ap.add_argument("--mem-fraction", type=float, default=0.85,
                    help="share of the free HBM the resident sessions may take (caps --concurrent)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()
        rank = dist.get_rank()

    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(local_rank)

    import numpy as np
    # every proof in flight keeps its own resident assignment (93 tables of 2^bn elements) plus scratch
    free_b, _total_b = gk.mem_info()
    per_session = (94.25 if args.circuit == "mimc" else 104) * 32 * (1 << args.bn)   # 93 tables + two half-size scratch tables + pyramids
    nconc = max(1, min(args.concurrent, args.steps, int(args.mem_fraction * free_b // per_session)))
    if nconc > 1 and args.steps % nconc and args.steps % (nconc - 1) == 0:
        nconc -= 1                                    # K steps deal evenly to one lane fewer: no straggler lane
    if dist is not None:
        t = torch.tensor([nconc], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)     # same number of lanes on every rank
        nconc = int(t.item())
    if dist is not None:
        # install the library's own RCCL communicators, one per lane (the per-round all-reduce of the limb-split
        # sums lives inside the C++ round loop); torch.distributed only carries the 128-byte unique ids, the
        # barriers and the max-over-ranks of the timing
        one_node = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world
        use_shm = args.exchange == "shm" or (args.exchange == "auto" and one_node)
        tag = [("%d_%d" % (os.getpid(), int(time.time() * 1e3))) if rank == 0 else None]
        dist.broadcast_object_list(tag, src=0)        # a name no earlier run can have left behind
        shm_name = "/gkrhip_bench_%s" % tag[0]
        transport, err = "rccl", ""
        if use_shm:
            # 576 bytes per round: the round kernel hands its sums to the host as in the un-sharded case and the ranks add
            # them through shared memory -- no collective kernel has to queue behind the compute-bound rounds
            gk.comm_init_shm_lanes(world, rank, nconc, shm_name)
            transport = "host shared memory (one node)"
        else:
            try:
                box = [b"".join(gk.comm_unique_id().tobytes() for _ in range(nconc)) if rank == 0 else None]
            except Exception as e:      # noqa: BLE001 -- reported below, never silent
                box, err = [None], str(e)
            dist.broadcast_object_list(box, src=0)
            ok = 0
            if box[0] is not None:
                try:
                    gk.comm_init_lanes(world, rank, np.frombuffer(box[0], dtype=np.uint8).copy().reshape(nconc, 128))
                    ok = 1
                except Exception as e:  # noqa: BLE001
                    err = str(e)
            t = torch.tensor([ok], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                # RCCL communicators could not be created on some rank: the same call sites run over the library's
                # host shared-memory transport (single node only) and the JSON line says so
                gk.comm_destroy()
                dist.barrier()
                gk.comm_init_shm_lanes(world, rank, nconc, shm_name)
                transport = "host shared memory (RCCL communicator init failed: %s)" % (err or "on another rank")
                if rank == 0:
                    print("bench.py: " + transport, file=sys.stderr)
    gamma = (world.bit_length() - 1) if dist is not None else 0
    # weak scaling: every GPU holds a 2^bn shard, the job proves 2^(bn + log2 N) hashes in ONE proof
    bn = args.bn + gamma
    import threading
    sessions = []
    layers = gk.gmimc_t2_circuit() if args.circuit == "gmimc" else None
    for _ in range(nconc):
        s = gk.MimcSession(bn, layers=layers)
        s.synth_inputs()        # block = initstate = RandomFrArray(2^bN), generated in HBM
        s.assign()              # Circuit.Assign: outside the timer, as BenchmarkGkr
        sessions.append(s)
    qprime = random_fr_array_np(bn)   # qPrime = RandomFrArray(bN), as gkr/gkr_test.go:93-95
    last = [None] * nconc

    def run_steps(total):
        """`total` full proofs; with nconc > 1 they are dealt round-robin to nconc sessions that prove
        concurrently (one host thread each; ctypes releases the GIL inside the library)."""
        if nconc == 1:
            for _ in range(total):
                last[0] = sessions[0].prove(qprime)
            return
        counts = [total // nconc + (1 if k < total % nconc else 0) for k in range(nconc)]

        def work(k):
            for _ in range(counts[k]):
                last[k] = sessions[k].prove(qprime)

        ths = [threading.Thread(target=work, args=(k,)) for k in range(nconc)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    def sync_all():
        gk.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    run_steps(max(args.warmup, 1 if nconc > 1 and args.warmup else 0))
    # single-proof latency (one proof alone on the GPU), reported beside the throughput figure
    gk.profile_reset(1 << args.bn)
    sync_all()
    tl = time.perf_counter()
    last[0] = sessions[0].prove(qprime)
    sync_all()
    latency_ms = 1e3 * (time.perf_counter() - tl)
    solo = gk.profile_get()          # the same launches with no other proof in flight
    gk.profile_reset(1 << args.bn)   # HIP-event accounting of the round-0 fold / partial-eval launches
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    flat = last[0]
    prof = gk.profile_get()
    gk.profile_reset(0)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    hashes = float(1 << bn) * args.steps
    out = {
        "metric": ("MiMC hashes GKR-proved/sec at bN=%d" if args.circuit == "mimc" else
                   "GMiMC(t=2) compressions GKR-proved/sec at bN=%d") % bn,
        "value": hashes / dt,
        "unit": "hashes/s",
        "n_gpus": world if dist is not None else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32x8 (BN254-Fr Montgomery, 256-bit integer)",
        "data": "synthetic",
        "config": {"workload": "gkr.Prove(MimcCircuit): ONE proof of 2^%d MiMC hashes, hypercube sharded on its low "
                               "index bits over %d GPU(s) (2^%d-entry shard per GPU), inputs RandomFrArray, "
                               "assignment resident in HBM" % (bn, max(world, 1) if dist is not None else 1, args.bn),
                   "bN": bn, "bN_per_gpu": args.bn, "proof_elements": int(flat.shape[0]),
                   "concurrent_proofs": nconc, "single_proof_latency_ms": latency_ms},
    }
    if dist is not None:
        out["config"]["per_round_exchange"] = transport + ": all-reduce of 72 limb-split u64 lanes per sumcheck round"
    if solo["fold_launches"]:
        # The fold launches on full-size tables, timed with HIP events on the launching stream.  Primary figure:
        # the single-proof pass (one proof alone on the GPU) that bench.py runs between the warm-up and the K timed
        # steps -- the kernel's own speed.  Inside the K timed steps the other lanes' VALU-bound kernels hold the
        # CUs, so an HBM-bound launch mostly waits for CU slots; that figure is reported under in_timed_region.
        traffic = None
        try:   # PMC pass of this very workload (tools/pmc_bench.sh), committed under profiles/
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fold_traffic.json")))
            if pm.get("bn") == args.bn:
                traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        sms = solo["fold_ms"] / solo["fold_launches"]
        sb = solo["fold_bytes"] / solo["fold_launches"]
        ach = sb / (sms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_fold (round-0 instance fold, 2^%d-element tables)" % args.bn,
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "launches": solo["fold_launches"], "avg_launch_ms": sms,
                           "algorithmic_bytes_per_launch": sb,
                           "measured": "HIP events on the launching stream around the fold launches of full-size tables "
                                       "during the single-proof pass of this run (one proof alone on the GPU)"}
        if prof["fold_launches"]:
            avg_ms = prof["fold_ms"] / prof["fold_launches"]
            bpl = prof["fold_bytes"] / prof["fold_launches"]
            out["roofline"]["in_timed_region"] = {
                "achieved": bpl / (avg_ms * 1e-3) / 1e9, "frac": bpl / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "avg_launch_ms": avg_ms, "launches": prof["fold_launches"],
                "measured": "same launches and events inside the K timed steps with %d proofs in flight: the launch "
                            "duration includes waiting for CU slots held by the other lanes' VALU-bound kernels" % nconc}
    if rank == 0:
        # the same kernel alone on the GPU (micro-benchmark of BenchmarkFolding's shape, poly/multilin_test.go:55-78)
        ms3 = gk.bench_fold(1 << args.bn, ntab=3, warmup=2, iters=10)
        out["fold_alone"] = {"tables": 3, "elements_per_table": 1 << args.bn, "ms": ms3,
                             "GB_per_s": 96.0 * 3 * (1 << (args.bn - 1)) / (ms3 * 1e-3) / 1e9,
                             "note": "device-resident micro-benchmark of the same kernel (gkrhip_bench_fold): three "
                                     "tables, no other kernel running"}
    if solo["peval_launches"]:
        # the dominant kernel by time is VALU-bound (exact 256-bit modular arithmetic: no MFMA, no HBM limit) and runs
        # at the vector issue rate, so it is priced against an instruction-issue ceiling: the round-0 loop body is
        # 3903 vector instructions per index pair (3566 half-rate: v_mad_u64_u32 / carries at 4.3 cycles per wave, 337
        # at 2.4; tools/isa_loop_count.py, rates from profiles/r01_ubench_*.txt) = 16143 issue cycles per pair and
        # wave; 1024 SIMDs x 64 lanes share the pairs; nominal clock 2.4 GHz (the sustained clock under this load is
        # nearer 2.1 GHz, profiles/r01_v8_pmc_round_kernel_sq.json)
        pairs = float(1 << (args.bn - 1))
        issue_cycles_per_pair = 16143.0
        ceiling_ms = pairs / (1024 * 64) * issue_cycles_per_pair / 2.4e9 * 1e3
        avg_ms = solo["peval_ms"] / solo["peval_launches"]
        out["partial_eval"] = {"kernel": "k_cipher_round_wide (round 0 of a cipher layer: 2^%d index pairs; per pair 10 field products, "
                                         "2 of them by a launch-wide constant, and 7 multiply-accumulates with deferred "
                                         "reduction)" % (args.bn - 1),
                               "bound": "integer VALU issue (no MFMA: modular arithmetic)",
                               "launches": solo["peval_launches"], "avg_launch_ms": avg_ms,
                               "ceiling_ms": ceiling_ms, "frac": ceiling_ms / avg_ms,
                               "vector_instructions_per_pair": 3903, "issue_cycles_per_pair": issue_cycles_per_pair,
                               "field_products_per_s": solo["peval_modmuls"] / (solo["peval_ms"] * 1e-3),
                               "ceiling_assumption": "every vector instruction of the loop body at its measured issue cost "
                                                     "(4.3 / 2.4 cycles per wave), two waves per SIMD keeping the port "
                                                     "busy, 2.4 GHz",
                               "measured": "HIP events around the round-0 launches of the single-proof pass"}
        if args.circuit != "mimc":       # the launch mix of other circuits differs (linear layers): no ceiling claimed
            for k in ("ceiling_ms", "frac", "vector_instructions_per_pair", "issue_cycles_per_pair", "ceiling_assumption"):
                out["partial_eval"].pop(k, None)
            out["partial_eval"]["kernel"] = "round-0 launches of the circuit's layers (cipher and linear)"
        if prof["peval_launches"]:
            out["partial_eval"]["in_timed_region"] = {
                "launches": prof["peval_launches"], "avg_launch_ms": prof["peval_ms"] / prof["peval_launches"],
                "field_products_per_s_per_launch": prof["peval_modmuls"] / (prof["peval_ms"] * 1e-3),
                "note": "launch durations with %d proofs in flight overlap the other lanes' kernels" % nconc}
    if prof.get("rounds"):
        out["host_split_ms_per_step"] = {k: prof[k] / args.steps for k in
                                         ("host_hash_ms", "host_wait_ms", "host_launch_ms", "host_other_ms")}
        out["host_split_ms_per_step"]["rounds"] = prof["rounds"] / args.steps
    if rank == 0 and not args.no_cpu_baseline and (dist is None or world == 1) and args.circuit == "mimc":
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    for s in sessions:
        s.close()
    if dist is not None:
        gk.comm_destroy()
        dist.destroy_process_group()


def random_fr_array_np(n):
    """common.RandomFrArray(n) as Montgomery limbs, computed with Python ints (host logic, tiny)."""
    import numpy as np
    Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    R = (1 << 256) % Q
    out = np.zeros((n, 4), np.uint64)
    for i in range(n):
        m = ((((i * i) & 0xFFFFFFFFFFFFFFFF) ^ 0xF45C9DF123F) % Q) * R % Q
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


if __name__ == "__main__":
    main()
