"""Build libgkrhip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot).

The SHA-256 of the sources and flags is compiled INTO the library (-DGKRHIP_SOURCE_SHA, exported by gkrhip_build_id() and
findable in the file as "GKRHIP_SOURCE_SHA=<hex>"): the loader and needs_build() compare the binary itself with the
sources on disk, so a stale git-ignored .so next to freshly pulled sources (whose tracked build_info.json already
describes the new sources) is rebuilt / refused instead of being called through a changed ABI.  build_info.json holds
what else the build knows about itself: the instruction counts of the round kernels' main loops and the register / LDS /
scratch figures of every kernel, taken from the ISA of this very build (bench.py prices the VALU-bound kernel against
them; nothing is hard-coded there)."""
import collections
import hashlib
import json
import os
import re
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgkrhip.so")
INFO = os.path.join(HERE, "build_info.json")
SRC = os.path.join(HERE, "csrc", "gkrhip.hip")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
              if f.endswith((".hip", ".h", ".inc"))) + [os.path.join(os.path.dirname(HERE), "include", "gkrhip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xarch_host", "-mbmi2", "-Xarch_host", "-madx"] + \
        os.environ.get("GKRHIP_EXTRA_FLAGS", "").split()      # experiments (e.g. -DGKR_WIDE_MUL2); part of the recorded build

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


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # host side: BMI2/ADX (mulx, adcx/adox) for the Fiat-Shamir hash's dependent multiplication chain (-16..20 % latency);
    # gkrhip_init refuses a CPU without them
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "libgkrhip.so")
        cmd = [hipcc] + FLAGS + ['-DGKRHIP_SOURCE_SHA="%s"' % source_sha(), "-save-temps", "-o", out, SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=tmp, stdout=None if verbose else subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        asm_text = open(os.path.join(tmp, asm[0])).read() if asm else ""
        counts = loop_counts(asm_text)
        resources = kernel_resources(asm_text)
        shutil.move(out, SO)
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()
    info = {"source_sha256": source_sha(), "flags": FLAGS, "hipcc": ver[0] if ver else "",
            "round_kernel_loops": counts, "kernel_resources": resources,
            "note": "round_kernel_loops: instructions of one index pair's loop body in the ISA of this build "
                    "(half_rate: v_mad_u64_u32, carries, v_mul_lo/hi, 64-bit shifts/adds; full_rate: the other vector instructions)"}
    json.dump(info, open(INFO, "w"), indent=1, sort_keys=True)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
    print(json.dumps(read_info()["round_kernel_loops"], indent=1))
