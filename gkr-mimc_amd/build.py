"""Build libgkrhip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot).

Several translation units, compiled side by side (UNITS below): gkrhip.hip -- all host code and the small kernels -- and one unit
per group of heavy template kernels (csrc/kern_unit.hip with -DGKR_GROUP_<NAME>; the instantiations are listed in
csrc/kernel_groups.h and declared `extern template` in gkrhip.hip).  A unit is recompiled only when a file it includes changed
(dependency files of the last compilation, objects kept under gkr-mimc_amd/build/, git-ignored): an edit of one kernel header
costs the units that include it, in parallel, not the whole library.  The link refuses undefined symbols.

The SHA-256 of the sources and flags is compiled INTO the library (csrc/build_id.cpp, -DGKRHIP_SOURCE_SHA, exported by
gkrhip_build_id() and findable in the file as "GKRHIP_SOURCE_SHA=<hex>"): the loader and needs_build() compare the binary itself
with the sources on disk, so a stale git-ignored .so next to freshly pulled sources (whose tracked build_info.json already
describes the new sources) is rebuilt / refused instead of being called through a changed ABI.  build_info.json holds
what else the build knows about itself: the instruction counts of the round kernels' main loops and the register / LDS /
scratch figures of every kernel, taken from the ISA of this very build (bench.py prices the VALU-bound kernel against
them; nothing is hard-coded there)."""
import collections
import concurrent.futures
import hashlib
import json
import os
import re
import shutil
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgkrhip.so")
INFO = os.path.join(HERE, "build_info.json")
CSRC = os.path.join(HERE, "csrc")
WORK = os.path.join(HERE, "build")          # objects, dependency files and the ISA of the last compilation of every unit
SRC = os.path.join(CSRC, "gkrhip.hip")
DEPS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
              if f.endswith((".hip", ".h", ".inc", ".cpp"))) + [os.path.join(os.path.dirname(HERE), "include", "gkrhip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xarch_host", "-mbmi2", "-Xarch_host", "-madx"] + \
         os.environ.get("GKRHIP_EXTRA_FLAGS", "").split()      # experiments (e.g. -DGKR_WIDE_MUL2); part of the recorded build
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-ldl", "-lpthread"]
FLAGS = CFLAGS + LDFLAGS                     # (what the recorded hash covers)
# (name, source, extra flags): gkrhip.hip first -- the longest single compilation starts first
UNITS = [("host", "gkrhip.hip", [])] + \
        [(g.lower(), "kern_unit.hip", ["-DGKR_GROUP_" + g]) for g in ("MSM_G2A", "MSM_G2B", "MSM_G2C", "MSM_G1", "WIDE2", "WIDEPRE", "ROUND", "NTT")]

# half-rate vector instructions on gfx950 (4.2-4.4 cycles per wave: profiles/r01_ubench_*.txt); the others issue in 2.4
HALF = ("v_mad_u64", "v_addc", "v_subb", "v_add_co", "v_sub_co", "v_subrev_co", "v_mul_lo", "v_mul_hi", "v_lshl_add_u64",
        "v_lshrrev_b64", "v_lshlrev_b64", "v_alignbit", "v_fma_f64", "v_add_f64", "v_mul_f64")
LOOP_KERNELS = {"round0": "k_cipher_round_wideILb0ELb1ELb0E", "fold_late": "k_cipher_round_wideILb1ELb1ELb0E",
                "fold_early": "k_cipher_round_wideILb1ELb0ELb0E", "round0_pre": "k_cipher_round_wideILb0ELb1ELb1E",
                # the bucket accumulation of the MSM: its INNERMOST loop is one mixed addition in the common case (the first
                # point of a bucket and the doubling / cancellation cases leave it: g1.hip.h)
                "msm_accumulate": "k_msm_accumulateI3FpFE",
                # computeH's tile kernels: the innermost loop is one sub-pass of two stages on a lane's four elements
                # (four butterflies, their LDS traffic and twiddle loads)
                "ntt_dif": "k_ntt_tileILb0ELb1EE", "ntt_dit": "k_ntt_tileILb1ELb0EE"}
INNERMOST = ("msm_accumulate", "ntt_dif", "ntt_dit")


def source_sha():
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def loop_counts(asm_text):
    """Instruction counts of the largest loop (label .. backward branch) of each round kernel."""
    out = {}
    for key, pat in LOOP_KERNELS.items():
        names = [n for n in re.findall(r"^(_Z\w+):", asm_text, re.M) if pat in n]
        if not names:
            continue
        name = names[0]
        i = asm_text.index(name + ":")
        body = asm_text[i:asm_text.index(".Lfunc_end", i)].splitlines()
        labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        loops = []
        for n, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                if any("s_endpgm" in x for x in body[labels[m.group(1)]:n]):
                    continue                       # a jump back to an exit block (early return), not a loop
                loops.append((labels[m.group(1)], n))
        if key in INNERMOST:                       # the largest loop that contains no other loop
            loops = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        for lp in loops:
            if best is None or lp[1] - lp[0] > best[1] - best[0]:
                best = lp
        if best is None:
            continue
        cnt = collections.Counter()
        for l in body[best[0]:best[1]]:
            l = l.strip()
            if l and not l.startswith((".", ";", "//")) and not l.endswith(":"):
                cnt[l.split()[0]] += 1
        valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
        half = sum(v for k, v in cnt.items() if k.startswith(HALF))
        out[key] = {"kernel": name, "loop_instructions": sum(cnt.values()), "vector": valu, "half_rate": half,
                    "full_rate": valu - half, "v_mad_u64_u32": cnt.get("v_mad_u64_u32", 0),
                    "carry": sum(v for k, v in cnt.items() if k.startswith(("v_addc", "v_subb", "v_add_co", "v_sub_co", "v_subrev_co"))),
                    "v_mul_lo_u32": cnt.get("v_mul_lo_u32", 0)}
    return out


def kernel_resources(asm_text):
    """VGPRs, scratch and LDS of every kernel of this build (from the .amdhsa_ directives of the ISA)."""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm_text, re.S):
        name, body = m.group(1), m.group(2)

        def val(key):
            mm = re.search(r"\.amdhsa_%s\s+(\d+)" % key, body)
            return int(mm.group(1)) if mm else None
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        short = re.sub(r"\(.*", "", short).replace("void ", "")
        out[short] = {"vgpr": val("next_free_vgpr"), "accum_offset": val("accum_offset"),
                      "scratch_bytes": val("private_segment_fixed_size"), "lds_bytes": val("group_segment_fixed_size")}
    return out


def read_info():
    """build_info.json, or None when it does not describe the library that is on disk."""
    try:
        info = json.load(open(INFO))
    except Exception:
        return None
    return info if info.get("source_sha256") == binary_sha() else None


def binary_sha(path=None):
    """The source hash compiled into the library file (None: no library, or one built before the hash was embedded)."""
    try:
        data = open(path or SO, "rb").read()
    except OSError:
        return None
    m = re.search(rb"GKRHIP_SOURCE_SHA=([0-9a-f]{64})", data)
    return m.group(1).decode() if m else None


def needs_build():
    return binary_sha() != source_sha()


def _unit_deps(name):
    """The files unit `name` included when it was last compiled (its dependency file), or None."""
    try:
        txt = open(os.path.join(WORK, name, "unit.d")).read()
    except OSError:
        return None
    files = [f for f in txt.replace("\\\n", " ").split(":", 1)[1].split() if f.startswith(os.path.dirname(HERE))]
    return sorted(set(files))


def _unit_key(name, src, extra, deps):
    h = hashlib.sha256()
    h.update(" ".join(CFLAGS + extra + [src]).encode())
    for d in deps:
        try:
            h.update(os.path.basename(d).encode() + b"\0" + open(d, "rb").read())
        except OSError:
            return None
    return h.hexdigest()


def _compile_unit(hipcc, name, src, extra, force, verbose):
    """Compile one unit unless its object is current; returns (name, seconds, rebuilt)."""
    wd = os.path.join(WORK, name)
    obj, keyf = os.path.join(wd, "unit.o"), os.path.join(wd, "unit.key")
    deps = _unit_deps(name)
    if not force and deps is not None and os.path.exists(obj):
        try:
            if open(keyf).read() == _unit_key(name, src, extra, deps):
                return name, 0.0, False
        except OSError:
            pass
    shutil.rmtree(wd, ignore_errors=True)
    os.makedirs(wd)
    cmd = [hipcc] + CFLAGS + extra + ["-c", "-MD", "-MF", "unit.d", "-save-temps", "-o", "unit.o", os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.time()
    subprocess.check_call(cmd, cwd=wd, stdout=None if verbose else subprocess.DEVNULL)
    for f in os.listdir(wd):            # keep the object, the dependency file and the device ISA; drop the other temporaries
        if not (f in ("unit.o", "unit.d") or f.endswith("gfx950.s")):
            os.remove(os.path.join(wd, f))
    open(keyf, "w").write(_unit_key(name, src, extra, _unit_deps(name)))
    return name, time.time() - t0, True


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # host side: BMI2/ADX (mulx, adcx/adox) for the Fiat-Shamir hash's dependent multiplication chain (-16..20 % latency);
    # gkrhip_init refuses a CPU without them
    os.makedirs(WORK, exist_ok=True)
    jobs = int(os.environ.get("GKRHIP_BUILD_JOBS", "0")) or min(len(UNITS), os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        done = list(ex.map(lambda u: _compile_unit(hipcc, u[0], u[1], u[2], force, verbose), UNITS))
    if verbose:
        print("units: " + ", ".join("%s %.0f s" % (n, s) if r else "%s (current)" % n for n, s, r in done), flush=True)
    idobj = os.path.join(WORK, "build_id.o")
    subprocess.check_call(["g++", "-O1", "-fPIC", '-DGKRHIP_SOURCE_SHA="%s"' % source_sha(), "-c", os.path.join(CSRC, "build_id.cpp"), "-o", idobj])
    out = os.path.join(WORK, "libgkrhip.so")
    cmd = [hipcc] + [os.path.join(WORK, u[0], "unit.o") for u in UNITS] + [idobj] + LDFLAGS + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    asm_text = ""
    for u in UNITS:
        for f in sorted(os.listdir(os.path.join(WORK, u[0]))):
            if f.endswith("gfx950.s"):
                asm_text += open(os.path.join(WORK, u[0], f)).read() + "\n"
    counts = loop_counts(asm_text)
    resources = kernel_resources(asm_text)
    shutil.move(out, SO)
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()
    info = {"source_sha256": source_sha(), "flags": FLAGS, "units": [u[0] for u in UNITS], "hipcc": ver[0] if ver else "",
            "round_kernel_loops": counts, "kernel_resources": resources,
            "note": "round_kernel_loops: instructions of one index pair's loop body in the ISA of this build "
                    "(half_rate: v_mad_u64_u32, carries, v_mul_lo/hi, 64-bit shifts/adds; full_rate: the other vector instructions)"}
    json.dump(info, open(INFO, "w"), indent=1, sort_keys=True)
    return SO


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True) if needs_build() or "--force" in sys.argv else print("libgkrhip.so is current")
    print(json.dumps(read_info()["round_kernel_loops"], indent=1))
