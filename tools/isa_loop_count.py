#!/usr/bin/env python3
"""VALU instruction count of the main loop of a kernel, from the gfx950 ISA hipcc emits:
     python tools/isa_loop_count.py [kernel-name-substring ...]
Compiles gkr-mimc_amd/csrc/gkrhip.hip with -save-temps into a temporary directory, finds each kernel's largest loop
(label .. backward branch) and counts its instructions; 'half-rate' = v_mad_u64_u32, carries, v_mul_lo_u32, 64-bit
shifts/adds, v_alignbit (4.2-4.4 cycles per wave on gfx950, profiles/r01_ubench_*.txt), the rest of the vector
instructions issue in 2.4.  bench.py's partial_eval ceiling uses the round-0 figure printed here."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = ("v_mad_u64", "v_addc", "v_subb", "v_add_co", "v_sub_co", "v_subrev_co", "v_mul_lo", "v_mul_hi", "v_lshl_add_u64",
        "v_lshrrev_b64", "v_lshlrev_b64", "v_alignbit")


def main():
    pats = sys.argv[1:] or ["k_cipher_round_wideILb0ELb1E", "k_cipher_round_wideILb1ELb1E", "k_cipher_round_wideILb1ELb0E"]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-save-temps", "-o", os.path.join(tmp, "lib.so"),
                               os.path.join(ROOT, "gkr-mimc_amd", "csrc", "gkrhip.hip")], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = open(os.path.join(tmp, "gkrhip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for name in re.findall(r"^(_Z\w+):", s, re.M):
        if not any(p in name for p in pats):
            continue
        i = s.index(name + ":")
        body = s[i:s.index(".Lfunc_end", i)].splitlines()
        labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for n, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                if best is None or n - labels[m.group(1)] > best[1] - best[0]:
                    best = (labels[m.group(1)], n)
        cnt = collections.Counter()
        for l in body[best[0]:best[1]]:
            l = l.strip()
            if l and not l.startswith((".", ";", "//")) and not l.endswith(":"):
                cnt[l.split()[0]] += 1
        valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
        half = sum(v for k, v in cnt.items() if k.startswith(HALF))
        print("%s: loop instructions %d, vector %d (half-rate %d, full-rate %d), issue cycles per pair at 4.3/2.4: %.0f"
              % (name, sum(cnt.values()), valu, half, valu - half, 4.3 * half + 2.4 * (valu - half)))


if __name__ == "__main__":
    main()
