"""Build libgkrhip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot).

The build also records what it was made from: build_info.json holds the SHA-256 of the sources (the loader refuses a
library whose sources have changed since -- the .so is git-ignored, so a stale one could otherwise travel to the GPU
box unnoticed) and the instruction counts of the round kernels' main loops taken from the ISA of this very build
(bench.py prices the VALU-bound kernel against them; nothing is hard-coded there)."""
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
LOOP_KERNELS = {"round0": "k_cipher_round_wideILb0ELb1E", "fold_late": "k_cipher_round_wideILb1ELb1E",
                "fold_early": "k_cipher_round_wideILb1ELb0E"}


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
        for n, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                if best is None or n - labels[m.group(1)] > best[1] - best[0]:
                    best = (labels[m.group(1)], n)
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


def read_info():
    try:
        return json.load(open(INFO))
    except Exception:
        return None


def needs_build():
    info = read_info()
    return not os.path.exists(SO) or info is None or info.get("source_sha256") != source_sha()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # host side: BMI2/ADX (mulx, adcx/adox) for the Fiat-Shamir hash's dependent multiplication chain (-16..20 % latency);
    # gkrhip_init refuses a CPU without them
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "libgkrhip.so")
        cmd = [hipcc] + FLAGS + ["-save-temps", "-o", out, SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=tmp, stdout=None if verbose else subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        counts = loop_counts(open(os.path.join(tmp, asm[0])).read()) if asm else {}
        shutil.move(out, SO)
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()
    info = {"source_sha256": source_sha(), "flags": FLAGS, "hipcc": ver[0] if ver else "",
            "round_kernel_loops": counts,
            "note": "round_kernel_loops: instructions of one index pair's loop body in the ISA of this build "
                    "(half_rate: v_mad_u64_u32, carries, v_mul_lo/hi, 64-bit shifts/adds; full_rate: the other vector instructions)"}
    json.dump(info, open(INFO, "w"), indent=1, sort_keys=True)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
    print(json.dumps(read_info()["round_kernel_loops"], indent=1))
