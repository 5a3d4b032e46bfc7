"""The bench.py output contract, checked on the committed line of the last GPU run (profiles/): every key the
driver reads is there, with the meaning BASELINE.json gives it."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_bn24_default.json")))
    assert files, "no committed bench line"
    return json.loads(open(files[-1]).read().strip().splitlines()[-1])


def test_bench_line_contract():
    d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and "bN=24" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - (1 << 24) * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6     # whole-job throughput
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1
    assert r["traffic"] is None or abs(r["traffic"] / r["algorithmic_bytes_per_launch"] - 1) < 0.05   # no wasted re-reads
    assert r.get("launches", 0) >= 10                                    # not a single-sample figure
    if "partial_eval" in d and "mad_issue_frac" in d["partial_eval"]:       # round 2: ceilings from the build's own ISA
        pe = d["partial_eval"]
        lp = pe["loop_instructions_per_pair"]
        assert lp["vector"] == lp["half_rate"] + lp["full_rate"] and lp["v_mad_u64_u32"] > 1000
        assert 0 < pe["mad_issue_frac"] < pe["frac"] <= 1
        for k in ("sumcheck_cipher_bn22", "sumcheck_multi_identity_91_bn22", "fold_2p25"):
            assert k in d["micro"], k
        assert d["single_proof"]["latency_ms"] > d["ms_per_step"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    if "configs" in d:      # round 3: BASELINE configs 2 and 5 and the partial-evaluation micro-benchmark in the driver-run line
        for key, bn in (("bn20", 20), ("gmimc_bn22", 22)):
            e = d["configs"][key]
            assert e["proof_verified_by_native_gkr_verify"] is True
            assert abs(e["hashes_per_s"] - (1 << bn) * 1e3 / e["ms_per_step"]) / e["hashes_per_s"] < 1e-6
            assert e["single_proof_ms"] > e["ms_per_step"] > 0
            hq = e.get("with_16_hardware_queues")      # end of round 3: the same config in a child process with GKRHIP_HW_QUEUES=16
            if hq and "error" not in hq:
                assert hq["hw_queues"] == 16 and hq["proof_verified_by_native_gkr_verify"] is True and hq["hashes_per_s"] > 0
        assert d["micro"]["partial_eval_bn15"]["us_per_dispatch"] > 0
        assert c["ns_per_field_mul_per_core"] <= 25, "the CPU baseline must be in the class of gnark-crypto's assembly"
        assert c["fr_mul_isolated_ns"]["nocarry_unrolled"]["independent"] > 0
        sp = d["single_proof"]
        assert sp["prelaunched_rounds"] > 0 and sp["lookahead_round0"] > 0 and sp["coop_rounds"] > 0
    if "msm_g1_2p22" in d.get("micro", {}):      # round 4: the Groth16 pieces of SURVEY 8 f4 in the driver-run line
        assert r["traffic"] is None or "not measured in this run" in r["traffic_source"]      # the PMC figure is labelled as the builder's
        for key, lg in (("msm_g1_2p20", 20), ("msm_g1_2p22", 22)):
            e = d["micro"][key]
            assert abs(e["points_per_s"] - (1 << lg) / (e["ms"] * 1e-3)) / e["points_per_s"] < 1e-6
            assert e["windows"] == -(-255 // e["window_bits"]) and abs(sum(e["phases_ms"].values()) - e["ms"]) < 0.1 * e["ms"]
            acc = e["accumulate"]
            lp = acc["loop_instructions_per_addition"]
            assert lp["v_mad_u64_u32"] == 8 * 128 + 2 * 100        # one mixed addition: 8 products + 2 squarings, nothing else in the loop
            assert lp["vector"] == lp["half_rate"] + lp["full_rate"] and 0 < acc["frac"] <= 1
            assert acc["mixed_additions"] == e["windows"] * (1 << lg)
        h = d["micro"]["compute_h_2p24"]
        assert h["passes"] == 9 and 0 < h["frac_of_hbm_peak"] < 1 and h["ms"] <= 16.0          # <= 16 ms at 2^24
        assert abs(h["field_products_per_s"] - h["field_products"] / (h["ms"] * 1e-3)) / h["field_products_per_s"] < 1e-6
        for key in ("bn20", "gmimc_bn22"):
            assert "hw_queues" in d["configs"][key]
        assert "spec_rounds" not in sp or sp["spec_rounds"] == 0       # bN = 24: the speculative rounds stay off by their size limit


def test_multi_rank_line_is_marked_when_no_rccl_pass_succeeded():
    """Two ranks on ONE GPU (profiles/r03_*_2ranks_on_one_gpu_*.json): RCCL cannot form a communicator there, so the line comes
    from the shared-memory pass -- and must say so in top-level fields a driver cannot miss."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*_bench_2ranks_on_one_gpu_bn22.json")))
    if not files:
        return
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["degraded"] == "rccl_failed" and d["n_gpus_rccl"] == 0 and d["value_transport"] == "shm"
    assert d["headline_pass"] == "shm" and {"shm", "rccl_one_lane", "rccl_tick"} <= set(d["passes"]) <= {"shm", "shm_tick", "rccl_one_lane", "rccl_tick", "rccl_tick_dev", "rccl_lanes"}
    assert all("error" in v for p, v in d["passes"].items() if p.startswith("rccl")) and d["passes"]["shm"]["value"] > 0


def test_bench_defaults_finish_quickly():
    """No flags = N=1 and a K/W that finish within minutes (10 steps of one 2^24-hash proof each, ~0.25 s per step)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_bench_rendezvous_over_tcp():
    """bench.py's torch-free bootstrap of a multi-rank pass (ids, barriers, min / max over the ranks) with three processes on
    127.0.0.1: broadcast of rank 0's object, min and max all-reduces, barriers, in the order a pass uses them."""
    import socket
    import subprocess
    import sys
    import textwrap
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import bench
        rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        r = bench.Rendezvous(rank, world, "127.0.0.1", port, timeout=60)
        assert r.allreduce(5 - rank, min) == 5 - (world - 1)
        blob = r.broadcast(b"x" * 384 if rank == 0 else None)
        assert blob == b"x" * 384
        assert r.allreduce(1 if rank != 1 else 0, min) == 0
        r.barrier()
        assert abs(r.allreduce(0.25 * (rank + 1), max) - 0.25 * world) < 1e-12
        r.barrier()
        r.close()
        print("RDV-OK", rank)
    """ % ROOT)
    ps = [subprocess.Popen([sys.executable, "-c", code, str(k), "3", str(port)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
          for k in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in ps]
    for k, (p, o) in enumerate(zip(ps, outs)):
        assert p.returncode == 0 and "RDV-OK %d" % k in o, o


def test_rendezvous_frames_are_data_and_strangers_are_ignored():
    """bench.py's torch-free bootstrap: frames are JSON -- nothing a peer sends is executed --, a connection
    without the run's token or with a rank that is out of range or already present is dropped, frame lengths are capped."""
    import importlib.util
    import socket
    import struct
    import threading
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    world, res, errs = 3, {}, []

    def run(r, delay):
        try:
            time.sleep(delay)
            d = bench.Rendezvous(r, world, "127.0.0.1", port, timeout=30)
            blob = bytes(range(256)) * 4
            res[r] = (d.allreduce(r + 1, max), d.allreduce(2.5 * r + 1, min), d.broadcast(blob if r == 0 else None),
                      d.broadcast("tag" if r == 0 else None), d.broadcast(None))
            d.barrier()
            d.close()
        except Exception as e:      # noqa: BLE001
            errs.append((r, repr(e)))

    ths = [threading.Thread(target=run, args=(r, 0.0 if r == 0 else 0.6)) for r in range(world)]
    for t in ths:
        t.start()
    time.sleep(0.2)       # rank 0 is listening; the real ranks arrive later
    for hello in (struct.pack("<i", 1) + b"x" * 16,          # right rank, wrong token
                  b"garbage",                                # short hello
                  struct.pack("<i", 99) + b"y" * 16):        # rank out of range
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=5) as c:
                c.sendall(hello)
                time.sleep(0.05)
        except OSError:
            pass
    for t in ths:
        t.join(60)
    assert not errs, errs
    assert all(res[r] == (3, 1.0, bytes(range(256)) * 4, "tag", None) for r in range(world)), res
    assert "pickle" not in open(os.path.join(ROOT, "bench.py")).read().split("class LocalRendezvous")[0].split("class Rendezvous")[1]


def test_supervisor_runs_a_dead_measuring_process_once_more(tmp_path):
    """bench.py at N = 1 measures in a child process; a child killed by a signal is started once more (an ordinary failure --
    here: no GPU in this container -- is not).  The parent never touches the GPU."""
    import importlib, subprocess, sys
    import pytest
    if importlib.import_module("gkr-mimc_amd").device_count() > 0:
        pytest.skip("a GPU is present: the full benchmark would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GKRHIP_BENCH_SELFTEST_ABORT_ONCE=str(tmp_path / "died"))
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert os.path.exists(tmp_path / "died")
    assert "died with code" in out.stderr and "running it once more" in out.stderr, out.stderr[-1500:]
    assert "no HIP device available" in out.stderr            # the second attempt ran (and failed the ordinary way: not retried)
    assert out.stderr.count("running it once more") == 1 and out.returncode != 0
    # without the injected death: one attempt only
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert "running it once more" not in out.stderr and out.returncode != 0


def test_supervisor_is_not_fooled_by_an_unrelated_preload(tmp_path):
    """The GPU boxes of the pool preload an exec guard into every process: an LD_PRELOAD that is not a profiler must not switch
    the supervisor off (it did until the end of round 5: the one abort in twelve was never retried); a profiler's must."""
    import importlib, subprocess, sys
    import pytest
    if importlib.import_module("gkr-mimc_amd").device_count() > 0:
        pytest.skip("a GPU is present: the full benchmark would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "guard.c"
    src.write_text("int gkrhip_test_guard_marker;\n")
    guard = str(tmp_path / "libguard.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", guard, str(src)])
    env = dict(os.environ, GKRHIP_BENCH_SELFTEST_ABORT_ONCE=str(tmp_path / "died"), LD_PRELOAD=guard)
    for k in [k for k in env if k.startswith(("ROCPROF", "ROCPROFILER_")) or k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIB", "ROCP_TOOL_LIBRARIES")]:
        env.pop(k)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert out.stderr.count("running it once more") == 1, out.stderr[-1500:]
    # a profiler's preload: the measurement runs in this process (which then dies of the injected abort, unsupervised)
    prof = str(tmp_path / "librocprofiler-sdk-tool.so")
    os.replace(guard, prof)
    env = dict(env, LD_PRELOAD=prof, GKRHIP_BENCH_SELFTEST_ABORT_ONCE=str(tmp_path / "died2"))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert "running it once more" not in out.stderr, out.stderr[-1500:]


def test_measuring_process_dies_with_the_supervisor_and_its_stderr_is_live(tmp_path):
    """a driver that kills bench.py (SIGTERM or SIGKILL) must not leave the measuring child behind with the GPU and its
    HBM; and the child's stderr reaches the log while it runs, not when it ends."""
    import signal, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sig in (signal.SIGTERM, signal.SIGKILL):
        pidfile = tmp_path / ("pid_%d" % int(sig))
        env = dict(os.environ, GKRHIP_BENCH_SELFTEST_HANG=str(pidfile))
        for k in [k for k in env if k.startswith(("ROCPROF", "ROCPROFILER_")) or k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIB", "ROCP_TOOL_LIBRARIES")]:
            env.pop(k)
        errlog = open(tmp_path / ("err_%d" % int(sig)), "wb")
        parent = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                                  stdout=subprocess.DEVNULL, stderr=errlog)
        t0 = time.time()
        while not (pidfile.exists() and pidfile.read_text().strip()) and time.time() - t0 < 120:
            time.sleep(0.1)
        child = int(pidfile.read_text())
        time.sleep(0.3)
        assert b"measuring process %d stalls" % child in open(errlog.name, "rb").read()      # live, while the child still runs
        os.kill(child, 0)                                                                     # (it does)
        parent.send_signal(sig)
        parent.wait(30)
        gone = False
        for _ in range(100):
            try:
                os.kill(child, 0)
                # a zombie of a reparented child counts as gone
                if open("/proc/%d/stat" % child).read().split(")")[-1].split()[0] == "Z":
                    gone = True
                    break
            except (ProcessLookupError, FileNotFoundError):
                gone = True
                break
            time.sleep(0.1)
        assert gone, "the measuring process survived its supervisor (signal %d)" % int(sig)


def test_config_summary_carries_the_second_tier_results():
    """the driver's record keeps `config` whole and drops `configs` / `micro` / `integrity`: the compact summary
    must name them all, and stay None-filled (not absent) when a part was skipped."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    empty = b.config_summary({})
    assert set(empty) == set(b.SUMMARY_KEYS) and empty["bn20"] is None and empty["bench_attempts"] == 1
    out = {"configs": {"bn20": {"hashes_per_s": 7.2e7, "single_proof_ms": 92.5, "concurrent_proofs": 72, "proofs_per_group": 3, "lanes_only": {"hashes_per_s": 6.4e7},
                                "proof_verified_by_native_gkr_verify": True},
                       "gmimc_bn22": {"hashes_per_s": 1.1e8, "single_proof_ms": 117.0, "concurrent_proofs": 12, "proof_verified_by_native_gkr_verify": True}},
           "micro": {"msm_g1_2p24": {"ms": 19.5}, "msm_g1_2p22": {"ms": 5.4}, "msm_g1_2p20": {"ms": 1.9}, "msm_g2_2p22": {"ms": 19.0},
                     "compute_h_2p24": {"ms": 12.7}, "msm_g1_fixed_base_2p24": {"ms": 18.0}},
           "oneshot_including_pcie": {"one_call_s": 0.31}, "roofline": {"frac": 0.79}, "partial_eval": {"frac": 0.86},
           "integrity": {"layer_checks": 920, "layer_check_failures": 0, "chal_retries": 0}}
    sm = b.config_summary(out)
    assert sm["bn20"] == {"hashes_per_s": 7.2e7, "single_proof_ms": 92.5, "lanes": 72, "proofs_per_group": 3, "lanes_only_hashes_per_s": 6.4e7, "single_calls_grouped_hashes_per_s": None, "verified": True}
    assert sm["gmimc_bn22"]["proofs_per_group"] == 1 and sm["gmimc_bn22"]["lanes_only_hashes_per_s"] is None
    assert sm["msm_g1_2p24_ms"] == 19.5 and sm["compute_h_2p24_ms"] == 12.7 and sm["oneshot_s"] == 0.31
    assert sm["msm_g1_fixed_base_ms"] == {"2p24": 18.0} and sm["layer_checks"] == 920 and sm["chal_retries"] == 0
    assert len(json.dumps(sm)) < 900          # compact: it must survive where the 2 000-character tail does not
    latest = _latest()
    if "summary" in latest["config"]:         # lines of round 6 on
        assert set(latest["config"]["summary"]) >= set(b.SUMMARY_KEYS) - {"msm_g1_fixed_base_ms", "msm_g2_fixed_base_2p22_ms"}
