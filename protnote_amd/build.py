"""Build libprotnote_hip.so (gfx950) in-tree with hipcc.  `python -m protnote_amd.build`.
Each translation unit is compiled to its own object (in parallel, only when stale), then linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
API = os.path.join(os.path.dirname(HERE), "include", "protnote_hip.h")
# translation unit -> headers it depends on
UNITS = {
    "protnote_hip.hip": ["gemm_engine.hpp", "gemm_bf16x3.hpp", "gemm_dma.hpp", "gemm_conv_dma.hpp", "gemm_conv_f64.hpp", "gemm_tn_fast.hpp", "gemm_tn.hpp", "train_kernels.hpp", "common.hpp"],
    "metrics.hip": ["common.hpp"],
}
SRC = [os.path.join(CSRC, u) for u in UNITS]
LIB = os.path.join(HERE, "libprotnote_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]


def _extra():
    return os.environ.get("PN_EXTRA_HIPCC_FLAGS", "").split()


def _flag_stamp():
    return " ".join(FLAGS + _extra())


def _obj(unit):
    return os.path.join(OBJ, unit.replace(".hip", ".o"))


def _unit_stale(unit) -> bool:
    o = _obj(unit)
    if not os.path.exists(o) or not os.path.exists(o + ".flags") or open(o + ".flags").read() != _flag_stamp():
        return True
    t = os.path.getmtime(o)
    deps = [os.path.join(CSRC, unit), API] + [os.path.join(CSRC, h) for h in UNITS[unit]]
    return any(os.path.getmtime(f) > t for f in deps)


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = set(SRC + [API])
    for u, hs in UNITS.items():
        deps.update(os.path.join(CSRC, h) for h in hs)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in deps)


def csrc_hash() -> str:
    """sha256 over the device sources (csrc/*.hip, csrc/*.hpp), file names included: profiles that describe kernel
    behaviour (profiles/*_hbm_traffic.json) are stamped with it so bench.py can tell when they no longer belong to the
    kernels it is running (.git does not travel to the GPU box)."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    for f in [os.path.join(CSRC, f) for f in files]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale() and not _extra():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(OBJ, exist_ok=True)

    def compile_unit(unit):
        if not force and not _unit_stale(unit):
            return
        cmd = [hipcc] + FLAGS + _extra() + ["-c", os.path.join(CSRC, unit), "-o", _obj(unit)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(_obj(unit) + ".flags", "w") as f:
            f.write(_flag_stamp())

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        list(ex.map(compile_unit, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
