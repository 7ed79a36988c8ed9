"""The stand-alone HIP tools of the round's kernel experiments stay buildable for gfx950 (cross-compiled here, no GPU needed):
tools/lab_bf16_nt.hip includes the product headers (csrc/gemm_bf16_m16.hpp, bwd_bf16_dz.hpp), so an interface change there that
breaks the kernel-level check shows up in the CPU suite."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,extra", [("tools/lab_bf16_nt.hip", ["-Iinclude", "-Iprotnote_amd/csrc", "-Itools", "-munsafe-fp-atomics"]),
                                       ("tools/mfma_power_probe.hip", [])])
def test_tool_compiles_for_gfx950(src, extra, tmp_path):
    out = tmp_path / "tool.o"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-value", "-Wno-comment", "--cuda-device-only", "-c"] + extra + [src, "-o", str(out)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.stat().st_size > 0
