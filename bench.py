#!/usr/bin/env python3
"""bench.py -- headline benchmark: MiMC hashes GKR-proved per second (gkr.Prove on
examples.MimcCircuit, assignment excluded from the timer as BenchmarkGkr does,
gkr/gkr_test.go:99-105), plus the fold kernel's achieved HBM bandwidth and the CPU oracle timed
beside it.

    python bench.py --gpus N --steps K --warmup W [--bn B] [--weak] [--exchange rccl|shm]

N = 1: one process, GPU 0, bN = 24 (BASELINE config 3, the size the metric is quoted on).
N > 1: launched by torch.distributed.run, one rank per GPU; ONE proof of 2^26 hashes (BASELINE config 4 at
N = 8: a 2^23-entry shard per GPU) sharded on the low index bits, the per-round sum of the round-polynomial
words all-reduced with RCCL (north_star's transport); on one node the same K steps run FIRST over the host
shared-memory exchange and are reported beside the RCCL figure -- and should the RCCL pass fail or stall (it has never run
on more than one GPU), the line is still printed, from the shared-memory pass, saying so.  --weak keeps 2^bn entries per
GPU instead (total 2^(bn + log2 N)).
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
NOMINAL_GHZ = 2.4         # MI355X engine clock (MI355X_MICROARCH.md)
HALF_RATE_CYCLES = 4.3    # v_mad_u64_u32 / carries / v_mul_lo_u32 per wave and SIMD (profiles/r01_ubench_instruction_rates.txt)
FULL_RATE_CYCLES = 2.4
N_SIMD = 1024             # 256 CUs x 4 SIMDs
BN_TOTAL_MULTI = 26       # BASELINE config 4


def cpu_baseline(target_seconds=40.0):
    """Time the CPU oracle (C restatement, OpenMP over the host cores) on a bounded sample of the same
    workload: gkr.Prove of 2^b MiMC hashes with RandomFrArray inputs; b grows until a run takes
    >= target_seconds/4 (each +1 doubles the work)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle
    # BASELINE config 1: bN = 10, the size of the reference's own CPU test path (gkr/gkr_test.go)
    i0 = coracle.random_fr_array(1 << 10)
    qp = coracle.random_fr_array(10)
    coracle.gkr_prove_mimc(10, i0, i0.copy(), qp, want_outputs=False)
    _, _, secs10 = coracle.gkr_prove_mimc(10, i0, i0.copy(), qp, want_outputs=False)
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
    cores = coracle.lib.oracle_num_threads()
    muls = 4555.0 * (1 << b)          # field multiplications of gkr.Prove per hash (SURVEY 8a totals)
    return {"value": (1 << b) / secs, "unit": "MiMC hashes GKR-proved/s", "cores": cores,
            "kind": "port", "ns_per_field_mul_per_core": secs * cores / muls * 1e9,
            "config1_bn10": {"seconds": secs10, "hashes_per_s": (1 << 10) / secs10},
            "sample": "gkr.Prove of 2^%d hashes (RandomFrArray inputs), %.2f s, C restatement of the reference "
                      "algorithm built with -O3 -march=x86-64-v3 -madx (portable unsigned __int128 CIOS product; not the Go "
                      "binary, whose gnark-crypto amd64 assembly is ~1.5-2x faster per multiplication)" % (b, secs)}


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


class ClockSampler:
    """Engine clock (sclk) of GPU `dev` while a region runs, from `rocm-smi --showclocks --json` in a side thread (a
    separate process each time: nothing of this process's HIP state is touched).  Best effort: empty if the tool is
    missing or prints something else."""

    def __init__(self, dev):
        self.dev, self.mhz, self._stop, self._t = dev, [], threading.Event(), None

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--json"], capture_output=True,
                                     text=True, timeout=5).stdout
                m = re.search(r'sclk[^()]*\((\d+)Mhz\)', out)
                if m:
                    self.mhz.append(int(m.group(1)))
            except Exception:   # noqa: BLE001
                return
            self._stop.wait(0.1)

    def __enter__(self):
        # under rocprofv3 the profiler's preloaded library initialises the GPU in every child as well, and rocm-smi
        # (an `env python3` script) would then exec from a GPU-initialised process: no sampling there
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
        if not profiled:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t is not None:
            self._t.join(timeout=10)

    def median(self):
        v = sorted(self.mhz)
        return v[len(v) // 2] if v else None


class Job:
    """nconc resident sessions (one lane each) of the same circuit and size, proving concurrently."""

    def __init__(self, gk, bn, nconc, layers):
        self.gk, self.bn, self.nconc = gk, bn, nconc
        self.sessions = []
        for _ in range(nconc):
            s = gk.MimcSession(bn, layers=layers)
            s.synth_inputs()        # block = initstate = RandomFrArray(2^bN), generated in HBM (this rank's shard)
            s.assign()              # Circuit.Assign: outside the timer, as BenchmarkGkr
            self.sessions.append(s)
        self.qprime = random_fr_array_np(bn)   # qPrime = RandomFrArray(bN), as gkr/gkr_test.go:93-95
        self.last = [None] * nconc
        self.errors = []

    def run_steps(self, total):
        """`total` full proofs; with nconc > 1 they are dealt round-robin to nconc sessions that prove
        concurrently (one host thread each; ctypes releases the GIL inside the library)."""
        if self.nconc == 1:
            for _ in range(total):
                self.last[0] = self.sessions[0].prove(self.qprime)
            return
        counts = [total // self.nconc + (1 if k < total % self.nconc else 0) for k in range(self.nconc)]

        def work(k):
            try:
                for _ in range(counts[k]):
                    self.last[k] = self.sessions[k].prove(self.qprime)
            except Exception as e:   # noqa: BLE001 -- re-raised on the main thread
                self.errors.append(e)

        ths = [threading.Thread(target=work, args=(k,)) for k in range(self.nconc)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if self.errors:
            raise self.errors[0]

    def close(self):
        for s in self.sessions:
            s.close()
        self.sessions = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bn", type=int, default=None,
                    help="log2 of the number of hashes of ONE proof: default 24 on one GPU (BASELINE config 3) and 26 in "
                         "total on N > 1 GPUs (config 4); with --weak it is the per-GPU shard size (default 23)")
    ap.add_argument("--weak", action="store_true", help="N > 1: keep 2^bn entries per GPU (total 2^(bn + log2 N))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true", help="skip the sumcheck / fold micro-benchmarks (SURVEY 8d)")
    ap.add_argument("--no-oneshot", action="store_true", help="skip the PCIe-inclusive one-shot calls from host buffers")
    ap.add_argument("--circuit", choices=["mimc", "gmimc"], default="mimc",
                    help="mimc: examples.MimcCircuit (the headline metric); gmimc: the build-defined GMiMC t=2 circuit "
                         "(BASELINE config 5, quoted at --bn 22)")
    ap.add_argument("--concurrent", type=int, default=5,
                    help="independent proofs in flight (each on its own resident session/lane/stream and, when "
                         "sharded, its own communicator); 1 = strictly one proof at a time")
    ap.add_argument("--exchange", choices=["rccl", "shm", "auto"], default="rccl",
                    help="transport of the per-round 576-byte sum of the sharded prover: rccl = ncclAllReduce over xGMI on "
                         "the lane's stream (the headline transport); shm = the ranks add the words on the host through "
                         "POSIX shared memory (one node only); auto = rccl")
    ap.add_argument("--no-shm-beside", action="store_true",
                    help="N > 1 on one node: do not repeat the timed steps over the shared-memory exchange")
    ap.add_argument("--device", type=int, default=None,
                    help="GPU ordinal of this rank (default LOCAL_RANK); several ranks on ONE GPU with --exchange shm is how "
                         "the multi-rank control flow is exercised on a single-GPU box")
    ap.add_argument("--mem-fraction", type=float, default=0.85,
                    help="share of the free HBM the resident sessions may take (caps --concurrent)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: start the ranks as children (one process per GPU, rendezvous on 127.0.0.1) and
        # pass their exit code on -- nothing in this process has touched the GPU
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (gloo's connection notes, RCCL's
    # version banner) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run
        # torch.distributed carries only the bootstrap (128-byte communicator ids), the barriers and the max-over-ranks
        # of the timing -- over gloo, so that the only RCCL in the process is the one libgkrhip dlopen()s and torch
        # never initialises the GPU
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        world = dist.get_world_size()
        rank = dist.get_rank()
    multi = dist is not None and world > 1
    if multi and args.exchange != "shm":
        # one hardware queue per lane stream (ROCclr's default is 4 queues for all streams): the collective kernels of
        # different lanes must be able to run side by side, whatever order the ranks issue them in
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    gk = importlib.import_module("gkr-mimc_amd")
    gk.init(local_rank if args.device is None else args.device)

    import numpy as np
    gamma = world.bit_length() - 1 if dist is not None else 0
    if dist is not None and (1 << gamma) != world:
        raise SystemExit("bench.py: the number of ranks must be a power of two")
    if not multi:
        bn = args.bn if args.bn is not None else 24
    elif args.weak:
        bn = (args.bn if args.bn is not None else 23) + gamma
    else:
        bn = args.bn if args.bn is not None else BN_TOTAL_MULTI
    bn_gpu = bn - gamma
    # every proof in flight keeps its own resident assignment (93 tables of 2^bn_gpu elements) plus scratch
    free_b, _total_b = gk.mem_info()
    per_session = (94.25 if args.circuit == "mimc" else 104) * 32 * (1 << bn_gpu)   # tables + two half-size scratch tables + pyramids
    if args.device is not None and dist is not None:
        free_b //= world                              # the ranks share one GPU
    nconc = max(1, min(args.concurrent, args.steps, int(args.mem_fraction * free_b // per_session)))
    if multi:
        nconc = min(nconc, 8)                         # one communicator per lane, at most 8
    if nconc > 1 and args.steps % nconc and args.steps % (nconc - 1) == 0:
        nconc -= 1                                    # K steps deal evenly to one lane fewer: no straggler lane
    if dist is not None:
        t = torch.tensor([nconc], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)     # same number of lanes on every rank
        nconc = int(t.item())
    layers = gk.gmimc_t2_circuit() if args.circuit == "gmimc" else None
    one_node = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world

    def install(transport_kind):
        """Install the library's communicators, one per lane (the per-round all-reduce of the limb-split sums lives
        inside the C++ round loop).  Returns a description of the transport actually installed."""
        tag = [("%d_%d" % (os.getpid(), int(time.time() * 1e3))) if rank == 0 else None]
        dist.broadcast_object_list(tag, src=0)        # a name no earlier run can have left behind
        shm_name = "/gkrhip_bench_%s" % tag[0]
        if transport_kind == "shm":
            gk.comm_init_shm_lanes(world, rank, nconc, shm_name)
            return "host shared memory (one node)"
        err = ""
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
        t = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            return "RCCL ncclAllReduce (ncclUint64, ncclSum) over xGMI, one communicator per lane"
        if not one_node:
            raise SystemExit("bench.py: RCCL communicator init failed (%s) and the ranks span several nodes" % err)
        # RCCL communicators could not be created on some rank: the same call sites run over the library's
        # host shared-memory transport (single node only) and the JSON line says so
        gk.comm_destroy()
        dist.barrier()
        gk.comm_init_shm_lanes(world, rank, nconc, shm_name)
        msg = "host shared memory (RCCL communicator init failed: %s)" % (err or "on another rank")
        if rank == 0:
            print("bench.py: " + msg, file=sys.stderr)
        return msg

    def sync_all():
        gk.synchronize()
        if dist is not None:
            dist.barrier()

    def timed(job, steps):
        sync_all()
        t0 = time.perf_counter()
        job.run_steps(steps)
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def run_phase(kind):
        """Sessions, warm-up, one proof alone (latency), the K timed steps, the native verifier -- over one transport
        (None: a single GPU).  Returns what the JSON line is assembled from; the job stays open."""
        ph = {"transport": install(kind) if dist is not None else None}
        job = Job(gk, bn, nconc, layers)
        ph["job"] = job
        job.run_steps(max(args.warmup, 1 if nconc > 1 and args.warmup else 0))
        # single-proof latency (one proof alone on the GPU), reported beside the throughput figure
        gk.profile_reset(1 << bn_gpu)
        sync_all()
        tl = time.perf_counter()
        job.last[0] = job.sessions[0].prove(job.qprime)
        sync_all()
        ph["latency_ms"] = 1e3 * (time.perf_counter() - tl)
        ph["solo"] = gk.profile_get()          # the same launches with no other proof in flight
        gk.profile_reset(1 << bn_gpu)          # HIP-event accounting of the round-0 fold / partial-eval launches
        with ClockSampler(local_rank if args.device is None else args.device) as clk:
            ph["dt"] = timed(job, args.steps)
        ph["clk"] = clk
        ph["flat"] = job.last[0]
        ph["prof"] = gk.profile_get()
        gk.profile_reset(0)
        # native gkr.Verify against the resident tables (outside the timer)
        ph["verified"] = bool(job.sessions[0].verify(job.qprime, ph["flat"]))
        return ph

    # N > 1 on one node with the RCCL headline: the SAME K steps run first over the host shared-memory exchange (the
    # transport every multi-rank test of this repository exercises), then over RCCL.  Should the RCCL pass fail or stall
    # (no multi-GPU box was available to the build), the line still appears: measured over the host exchange, saying so.
    shm_first = multi and one_node and args.exchange != "shm" and not args.no_shm_beside
    beside = None
    rccl_error = None
    if shm_first:
        beside = run_phase("shm")
        beside["job"].close()
        gk.comm_destroy()
        dist.barrier()
    watchdog = None
    emitted = threading.Event()
    if beside is not None:
        limit_s = float(os.environ.get("GKRHIP_BENCH_RCCL_LIMIT_S", "0")) or (120.0 + 20.0 * beside["dt"] * (1 + args.warmup / max(args.steps, 1)))

        def give_up():
            if emitted.is_set():
                return
            if rank == 0:
                emit(fallback_line("the RCCL pass did not finish within %.0f s" % limit_s))
            os._exit(0)           # collective kernels may still be spinning on the device: no orderly teardown

        watchdog = threading.Timer(limit_s, give_up)
        watchdog.daemon = True

    def fallback_line(why):
        """The JSON line from the shared-memory pass alone (the RCCL pass failed or stalled)."""
        o = {"metric": ("MiMC hashes GKR-proved/sec at bN=%d" if args.circuit == "mimc" else
                        "GMiMC(t=2) compressions GKR-proved/sec at bN=%d") % bn, "value": float(1 << bn) * args.steps / beside["dt"],
             "unit": "hashes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
             "ms_per_step": 1e3 * beside["dt"] / args.steps, "higher_is_better": True,
             "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
             "dtype": "u32x8 (BN254-Fr Montgomery, 256-bit integer)", "data": "synthetic",
             "config": {"workload": "gkr.Prove(MimcCircuit): ONE proof of 2^%d hashes (bN_total = %d), hypercube sharded on its "
                                    "low index bits over %d GPU(s) (2^%d-entry shard per GPU), inputs RandomFrArray, assignment "
                                    "resident in HBM; per-round exchange: %s" % (bn, bn, world, bn_gpu, beside["transport"]),
                        "bN": bn, "bN_total": bn, "bN_per_gpu": bn_gpu, "concurrent_proofs": nconc,
                        "single_proof_latency_ms": beside["latency_ms"],
                        "proof_verified_by_native_gkr_verify": beside["verified"],
                        "per_round_exchange": beside["transport"] + " (RCCL pass: " + why + ")"},
             "single_proof_latency_ms": beside["latency_ms"], "roofline": None, "cpu_baseline": None}
        return o

    try:
        if watchdog is not None:
            watchdog.start()
        head = run_phase(None if dist is None else ("shm" if args.exchange == "shm" else "rccl"))
    except Exception as e:      # noqa: BLE001 -- reported in the line
        if beside is None:
            raise
        rccl_error = str(e)
        emitted.set()
        if rank == 0:
            emit(fallback_line(rccl_error))
        os._exit(0)
    finally:
        if watchdog is not None:
            watchdog.cancel()
    emitted.set()
    transport, job, dt, latency_ms = head["transport"], head["job"], head["dt"], head["latency_ms"]
    solo, prof, clk, flat, verified = head["solo"], head["prof"], head["clk"], head["flat"], head["verified"]

    hashes = float(1 << bn) * args.steps
    n_gpus = world if dist is not None else 1
    out = {
        "metric": ("MiMC hashes GKR-proved/sec at bN=%d" if args.circuit == "mimc" else
                   "GMiMC(t=2) compressions GKR-proved/sec at bN=%d") % bn,
        "value": hashes / dt,
        "unit": "hashes/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak" if (args.weak or not multi) else "strong",
        "vs_baseline": None,
        "dtype": "u32x8 (BN254-Fr Montgomery, 256-bit integer)",
        "data": "synthetic",
        "config": {"workload": "gkr.Prove(%s): ONE proof of 2^%d hashes (bN_total = %d), hypercube sharded on its low "
                               "index bits over %d GPU(s) (2^%d-entry shard per GPU), inputs RandomFrArray, "
                               "assignment resident in HBM%s"
                               % ("MimcCircuit" if args.circuit == "mimc" else "GMiMC t=2 circuit", bn, bn, n_gpus, bn_gpu,
                                  ("; per-round exchange: " + transport) if transport else ""),
                   "bN": bn, "bN_total": bn, "bN_per_gpu": bn_gpu, "proof_elements": int(flat.shape[0]),
                   "concurrent_proofs": nconc, "single_proof_latency_ms": latency_ms,
                   "proof_verified_by_native_gkr_verify": verified},
        "single_proof_latency_ms": latency_ms,
    }
    if dist is not None:
        out["config"]["per_round_exchange"] = transport + ": all-reduce of 72 limb-split u64 lanes per sumcheck round"
        out["config"]["bootstrap"] = "torch.distributed gloo (ids, barriers, max-over-ranks of the timing); the only RCCL in the process is the one libgkrhip dlopen()s"
    if solo.get("rounds"):
        out["single_proof"] = {"latency_ms": latency_ms, "hashes_per_s": float(1 << bn) / (latency_ms * 1e-3),
                               "rounds": solo["rounds"], "host_hash_ms": solo["host_hash_ms"], "host_wait_ms": solo["host_wait_ms"],
                               "host_launch_ms": solo["host_launch_ms"], "host_other_ms": solo["host_other_ms"],
                               "note": "one gkr.Prove alone on the GPU (BenchmarkGkr's shape): the serial chain of rounds -- "
                                       "Fiat-Shamir hash on the host, then the next round kernel -- is not overlapped with anything; "
                                       "host_wait_ms is the time the host waited for round kernels (their GPU time plus hand-off latency)"}

    build_info = importlib.import_module("gkr-mimc_amd.build").read_info() or {}
    if rank == 0:
        # ---- roofline of the HBM-bound kernel: k_fold<1>, the launch gkr.Prove uses for full-size tables, timed with HIP
        # events on the launching stream over 20 back-to-back launches on 2^bn_gpu-element tables (out of place,
        # table[i] = Montgomery(i), r = 5: BenchmarkFolding's shape)
        iters = 20
        ms_b2b, ms1 = gk.bench_fold(1 << bn_gpu, ntab=1, warmup=3, iters=iters, isolated=True)
        bytes1 = 96.0 * (1 << (bn_gpu - 1))
        traffic = None
        try:   # PMC pass of the same launches (tools/pmc_bench.sh), committed under profiles/
            pm = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_fold_traffic.json")))
            if pm.get("bn") == bn_gpu:
                traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        ach = bytes1 / (ms_b2b * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_fold<1> on a 2^%d-element table (2^%d outputs)" % (bn_gpu, bn_gpu - 1),
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "launches": iters, "avg_launch_ms": ms_b2b,
                           "algorithmic_bytes_per_launch": bytes1,
                           "measured": "HIP events on the launching stream around %d launches queued back to back, nothing else "
                                       "running (gkrhip_bench_fold): wall time / %d.  rocprofv3's per-kernel average for the "
                                       "full-size launches agrees within 3 %% (profiles/r02_v6_solo_fold_launches_by_size.csv: "
                                       "127.6 us over 54 launches); 96 B per output element (SURVEY 8d)" % (iters, iters),
                           "one_at_a_time": {"avg_launch_ms": ms1, "achieved": bytes1 / (ms1 * 1e-3) / 1e9,
                                             "frac": bytes1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "measured": "the same %d launches one at a time on an idle GPU, one event pair each: "
                                                         "includes the launch latency and the clock ramp after every idle gap" % iters}}
        if solo["fold_launches"]:
            sms = solo["fold_ms"] / solo["fold_launches"]
            sb = solo["fold_bytes"] / solo["fold_launches"]
            out["roofline"]["in_single_proof"] = {
                "achieved": sb / (sms * 1e-3) / 1e9, "frac": sb / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": sms,
                "launches": solo["fold_launches"], "algorithmic_bytes_per_launch": sb,
                "measured": "the fold launches on full-size tables inside one gkr.Prove alone on the GPU (the key-copy "
                            "layer's round 0: one launch per table of the instance, timed together)"}
        if prof["fold_launches"]:
            avg_ms = prof["fold_ms"] / prof["fold_launches"]
            bpl = prof["fold_bytes"] / prof["fold_launches"]
            out["roofline"]["in_timed_region"] = {
                "achieved": bpl / (avg_ms * 1e-3) / 1e9, "frac": bpl / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "avg_launch_ms": avg_ms, "launches": prof["fold_launches"],
                "measured": "same launches inside the K timed steps with %d proofs in flight: the launch duration includes "
                            "waiting for CU slots held by the other lanes' VALU-bound kernels" % nconc}
    loops = build_info.get("round_kernel_loops", {})
    if solo["peval_launches"] and args.circuit == "mimc" and "round0" in loops:
        # the dominant kernel by time is VALU-bound (exact 256-bit modular arithmetic: no MFMA, no HBM limit): priced
        # against instruction-issue ceilings taken from the ISA of THIS build (gkr-mimc_amd/build_info.json)
        lp = loops["round0"]
        pairs = float(1 << (bn_gpu - 1))
        avg_ms = solo["peval_ms"] / solo["peval_launches"]
        waves_per_simd = pairs / (N_SIMD * 64)
        issue_cycles = HALF_RATE_CYCLES * lp["half_rate"] + FULL_RATE_CYCLES * lp["full_rate"]
        ceiling_ms = waves_per_simd * issue_cycles / (NOMINAL_GHZ * 1e9) * 1e3
        mad_ms = waves_per_simd * HALF_RATE_CYCLES * lp["v_mad_u64_u32"] / (NOMINAL_GHZ * 1e9) * 1e3
        # the algorithmic count, independent of the schedule: the round's 10 field products (schoolbook 64 limb products +
        # 64 of the Montgomery half each) and 7 plain multiply-accumulates (64) -- squarings, constant-multiplier images
        # and carry planning lower the instructions issued, not this number
        SCHOOLBOOK_LIMB_PRODUCTS = 10 * 128 + 7 * 64
        school_ms = waves_per_simd * HALF_RATE_CYCLES * SCHOOLBOOK_LIMB_PRODUCTS / (NOMINAL_GHZ * 1e9) * 1e3
        out["partial_eval"] = {
            "kernel": "k_cipher_round_wide<false,true> (round 0 of a cipher layer: 2^%d index pairs; per pair 10 field "
                      "products, 2 of them by a launch-wide constant, and 7 multiply-accumulates with deferred reduction)" % (bn_gpu - 1),
            "bound": "integer VALU issue (no MFMA: modular arithmetic)",
            "launches": solo["peval_launches"], "avg_launch_ms": avg_ms,
            "loop_instructions_per_pair": lp, "issue_cycles_per_pair": issue_cycles,
            "ceiling_ms": ceiling_ms, "frac": ceiling_ms / avg_ms,
            "mad_only_ms": mad_ms, "mad_issue_frac": mad_ms / avg_ms,
            "schoolbook_limb_products_per_pair": SCHOOLBOOK_LIMB_PRODUCTS, "schoolbook_frac": school_ms / avg_ms,
            "field_products_per_s": solo["peval_modmuls"] / (solo["peval_ms"] * 1e-3),
            "ceiling_assumption": "frac: every vector instruction of this build's loop body at its measured issue cost (%.1f / %.1f "
                                  "cycles per wave), the port never idle, nominal %.1f GHz; mad_issue_frac: the limb products "
                                  "(v_mad_u64_u32) alone at %.1f cycles -- what a carry-free multiplier would cost -- over the "
                                  "measured duration; schoolbook_frac: the same for the schoolbook limb products of the round's field "
                                  "operations (10 x 128 + 7 x 64 per pair), a count no schedule changes"
                                  % (HALF_RATE_CYCLES, FULL_RATE_CYCLES, NOMINAL_GHZ, HALF_RATE_CYCLES),
            "measured": "HIP events around the round-0 launches of the single-proof pass"}
        if clk.median():
            ghz = clk.median() * 1e-3
            out["partial_eval"]["sclk_mhz_during_timed_steps"] = {"median": clk.median(), "min": min(clk.mhz), "max": max(clk.mhz),
                                                                   "samples": len(clk.mhz), "source": "rocm-smi --showclocks"}
            out["partial_eval"]["frac_at_measured_clock"] = ceiling_ms * NOMINAL_GHZ / ghz / avg_ms
            out["partial_eval"]["mad_issue_frac_at_measured_clock"] = mad_ms * NOMINAL_GHZ / ghz / avg_ms
        if prof["peval_launches"]:
            out["partial_eval"]["in_timed_region"] = {
                "launches": prof["peval_launches"], "avg_launch_ms": prof["peval_ms"] / prof["peval_launches"],
                "note": "launch durations with %d proofs in flight overlap the other lanes' kernels" % nconc}
    if prof.get("rounds"):
        out["host_split_ms_per_step"] = {k: prof[k] / args.steps for k in
                                         ("host_hash_ms", "host_wait_ms", "host_launch_ms", "host_other_ms")}
        out["host_split_ms_per_step"]["rounds"] = prof["rounds"] / args.steps

    if beside is not None:   # the same K steps over the host shared-memory exchange (measured first), beside the RCCL headline
        out["config"]["shm_exchange_beside"] = {"value": hashes / beside["dt"], "ms_per_step": 1e3 * beside["dt"] / args.steps,
                                                "transport": beside["transport"], "single_proof_latency_ms": beside["latency_ms"]}

    if rank == 0 and not multi and not args.no_micro and args.circuit == "mimc":
        # SURVEY 8d micro-benchmarks, shaped like the reference's own (device-resident tables)
        job.close()
        micro = {}
        ms, _ = gk.bench_sumcheck(0, 22, 1, warmup=1, iters=3)
        micro["sumcheck_cipher_bn22"] = {"ms_per_prove": ms, "index_pairs_per_s": float(1 << 22) / (ms * 1e-3),
                                         "mirrors": "BenchmarkWithCipherGate, sumcheck/prover_test.go:96-109 (bn = 22, L = R = i, Ark = 145646)"}
        ms, _ = gk.bench_sumcheck(1, 22, 91, warmup=1, iters=3)
        micro["sumcheck_multi_identity_91_bn22"] = {"ms_per_prove": ms,
                                                    "mirrors": "BenchmarkMultiIdentity, sumcheck/prover_test.go:111-125 (bn = 22, 91 instances)"}
        ms = gk.bench_fold(1 << 25, ntab=1, warmup=2, iters=10)
        micro["fold_2p25"] = {"ms": ms, "GB_per_s": 96.0 * (1 << 24) / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": 96.0 * (1 << 24) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "mirrors": "BenchmarkFolding, poly/multilin_test.go:55-78 (2^25 elements, table[i] = i, r = 5)"}
        out["micro"] = micro
    if rank == 0 and not multi and not args.no_oneshot and args.circuit == "mimc":
        # the production caller's shape (GkrProverHint.Call, prover/gadget/hints.go:197-233): Assign + Prove from HOST
        # buffers in one call -- upload, limb-plane transposition, canonicality check, assignment, proof, download of
        # the output table.  PCIe-inclusive; never `value`.
        job.close()
        rng = np.random.default_rng(1)
        n = 1 << bn
        ins = []
        for _ in range(2):   # any canonical residues serve as inputs for a timing
            a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
            a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
            ins.append(a)
        qp = random_fr_array_np(bn)
        gk.gkr_prove_mimc(ins[0], ins[1], qp)                      # warm: arena, lane pool
        t0 = time.perf_counter()
        gk.gkr_prove_mimc(ins[0], ins[1], qp)
        one = time.perf_counter() - t0
        nthr = 4

        def call(_k):
            gk.gkr_prove_mimc(ins[0], ins[1], qp)

        ths = [threading.Thread(target=call, args=(k,)) for k in range(nthr)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        par = time.perf_counter() - t0
        out["oneshot_including_pcie"] = {
            "one_call_s": one, "one_call_hashes_per_s": n / one,
            "concurrent_calls": nthr, "concurrent_wall_s": par, "concurrent_hashes_per_s": nthr * n / par,
            "per_call_bytes_over_pcie": 3 * 32 * n,
            "note": "gkrhip_gkr_prove_mimc on pageable host buffers (2 x %d MiB up, %d MiB down per call); the concurrent figure is "
                    "%d calls from %d host threads, each on a lane of its own" % (32 * n >> 20, 32 * n >> 20, nthr, nthr)}
    if rank == 0 and not args.no_cpu_baseline and not multi and args.circuit == "mimc":
        out["cpu_baseline"] = cpu_baseline()
    out["build"] = {"source_sha256": (build_info.get("source_sha256") or "")[:16], "hipcc": build_info.get("hipcc", "")}
    if rank == 0:
        emit(out)
    job.close()
    if dist is not None:
        gk.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
