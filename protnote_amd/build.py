"""Build libprotnote_hip.so (gfx950) in-tree with hipcc.  `python -m protnote_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "protnote_hip.hip")]
HDR = [os.path.join(HERE, "csrc", "gemm_engine.hpp"),
       os.path.join(HERE, "csrc", "gemm_tn.hpp"),
       os.path.join(HERE, "csrc", "train_kernels.hpp"),
       os.path.join(os.path.dirname(HERE), "include", "protnote_hip.h")]
LIB = os.path.join(HERE, "libprotnote_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SRC + HDR)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    extra = os.environ.get("PN_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + ["-o", LIB] + SRC
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
