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
# translation units; EVERY unit depends on every csrc/*.hpp and on the API header (found by glob at call time - a hand-kept
# list once omitted two new headers and served a stale binary silently)
UNITS = ["protnote_hip.hip", "metrics.hip", "build_hash.cpp"]
SRC = [os.path.join(CSRC, u) for u in UNITS]
LIB = os.path.join(HERE, "libprotnote_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value"]
HASH_MARKER = b"PN_CSRC_HASH="


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))


def sources():
    """Every file the binary is made of: csrc/*.hip, csrc/*.hpp, csrc/*.cpp and the API header."""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))) + [API]


def _extra():
    return os.environ.get("PN_EXTRA_HIPCC_FLAGS", "").split()


def _unit_flags(unit):
    fl = FLAGS + _extra()
    if unit == "build_hash.cpp":  # the one unit that carries the source hash: recompiled (1 s) whenever any source changes
        fl = fl + [f'-DPN_CSRC_HASH="{csrc_hash()}"']
    return fl


def _obj(unit):
    return os.path.join(OBJ, os.path.splitext(unit)[0] + ".o")


def _unit_stamp(unit) -> str:
    """What an object file was made from: the compile flags and a sha256 over the unit, every csrc/*.hpp and the API header.
    Written beside the object (`<obj>.flags`) and compared by CONTENT: file times do not survive the snapshot that travels
    to the GPU box, so an object that merely looks newer than an edited source must not be relinked."""
    import hashlib

    h = hashlib.sha256()
    for f in [os.path.join(CSRC, unit), API] + headers():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return " ".join(_unit_flags(unit)) + "\nsrc=" + h.hexdigest()[:16]


def _unit_stale(unit) -> bool:
    o = _obj(unit)
    if not os.path.exists(o) or not os.path.exists(o + ".flags"):
        return True
    return open(o + ".flags").read() != _unit_stamp(unit)


def csrc_hash() -> str:
    """sha256 over every source of the binary (csrc/*.hip, *.hpp, *.cpp, include/protnote_hip.h), file names included.
    Compiled into the .so (pn_build_hash()); _lib.lib() refuses a binary whose hash differs from the sources beside it,
    and profiles that describe kernel behaviour (profiles/*_hbm_traffic.json) are stamped with it so bench.py can tell
    when they no longer belong to the kernels it is running (.git does not travel to the GPU box)."""
    import hashlib

    h = hashlib.sha256()
    for f in sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    # extra compile flags (PN_EXTRA_HIPCC_FLAGS: A/B builds of an experiment macro) are part of what the binary IS: a build
    # made with them is only accepted by a process that runs with the same flags in its environment
    h.update(" ".join(_extra()).encode())
    return h.hexdigest()[:16]


def embedded_hash(path: str = None):
    """The source hash stamped into a built .so, read from the file's bytes (no dlopen); None if there is none."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    blob = open(path, "rb").read()
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    return blob[i + len(HASH_MARKER): i + len(HASH_MARKER) + 16].decode("ascii", "replace")


def stale() -> bool:
    """True when the .so is missing or was not built from the sources now in csrc/ (content hash, not mtimes: the
    snapshot that travels to the GPU box does not keep them)."""
    return embedded_hash() != csrc_hash()


def have_hipcc():
    import shutil

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else shutil.which("hipcc")


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():  # (the hash covers PN_EXTRA_HIPCC_FLAGS)
        return LIB
    hipcc = have_hipcc()
    if not hipcc:
        raise RuntimeError("hipcc not found (set HIPCC)")
    os.makedirs(OBJ, exist_ok=True)
    # one builder at a time: the N ranks of `bench.py --gpus N` (or pytest-xdist workers) may all find the binary stale at once
    import fcntl

    lock = open(os.path.join(OBJ, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not stale():  # another process built it while this one waited
            return LIB
        return _build_locked(hipcc, force, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc, force, verbose):

    def compile_unit(unit):
        if not force and not _unit_stale(unit):
            return
        lang = ["-x", "hip"] if unit.endswith(".hip") else []
        cmd = [hipcc] + _unit_flags(unit) + lang + ["-c", os.path.join(CSRC, unit), "-o", _obj(unit)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        stamp = _unit_stamp(unit)  # taken BEFORE the compile: an edit during it leaves a stamp that no longer matches
        subprocess.check_call(cmd)
        with open(_obj(unit) + ".flags", "w") as f:
            f.write(stamp)

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        list(ex.map(compile_unit, UNITS))
    tmp = LIB + f".tmp{os.getpid()}"  # link beside the target, then rename: a reader never sees a half-written library
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    if embedded_hash() != csrc_hash():
        raise RuntimeError(f"built {LIB} carries hash {embedded_hash()} but the sources hash to {csrc_hash()} "
                           "(a source changed during the build?)")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
