"""The diagnostic shims and probes of tools/ still build in this image (they are not part of the product and nothing loads
them here: building the checker's tools is not using them).  gpu_efence.c and alloc_trace.c are what located the round-5
over-read (profiles/r05_anomalies.md (c)); the vmm_* probes are the stand-alone evidence for (d)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("src,extra", [("gpu_efence.c", ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-lpthread"]), ("alloc_trace.c", [])])
def test_preload_shims_build(tmp_path, src, extra):
    out = str(tmp_path / (src + ".so"))
    subprocess.check_call(["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tools", src), "-ldl"] + extra,
                          stderr=subprocess.DEVNULL)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", out], text=True)
    for name in ("hipMalloc", "hipFree"):
        assert (" T " + name) in syms, name


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", ["vmm_interior_probe.hip", "vmm_release_probe.hip", "vmm_reserve_limit.hip", "prio_event_probe.hip"])
def test_vmm_probes_build(tmp_path, src):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-o", str(tmp_path / "probe"), os.path.join(ROOT, "tools", src)],
                          stderr=subprocess.DEVNULL)
