"""Build libgkrhip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgkrhip.so")
SRC = os.path.join(HERE, "csrc", "gkrhip.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in
        ("gkrhip.hip", "host_ctx.hip.h", "host_coll.hip.h", "host_sumcheck.hip.h", "host_circuit.hip.h", "kernels.hip.h", "cipher_round.hip.h", "linear_round.hip.h", "fr_bn254.h", "fr_mont_gen.inc", "fr_mont2_gen.inc", "fr_mac_wide_gen.inc", "fr_mulc2_gen.inc", "fr_host.h", "arks_bn254.inc")] + \
       [os.path.join(os.path.dirname(HERE), "include", "gkrhip.h")]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # host side: BMI2/ADX (mulx, adcx/adox) for the Fiat-Shamir hash's dependent multiplication chain (-16..20 % latency);
    # gkrhip_init refuses a CPU without them
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Xarch_host", "-mbmi2", "-Xarch_host", "-madx",
           "-o", SO, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
