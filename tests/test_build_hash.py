"""The binary knows which sources it was built from, and the binding refuses any other (build.py's
hand-kept header list omitted two headers and `_lib.lib()` built only when the .so was MISSING, so an edited header
silently measured the old kernels).  All CPU: nothing here launches a kernel."""
import os
import shutil

import pytest

from protnote_amd import build


def _clone_sources(tmp_path):
    """A scratch copy of the package's source layout (csrc/ + include/) with the built .so beside it."""
    pkg = tmp_path / "protnote_amd"
    (pkg / "csrc").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    for f in os.listdir(build.CSRC):
        if f.endswith((".hip", ".hpp", ".cpp")):
            shutil.copy(os.path.join(build.CSRC, f), pkg / "csrc" / f)
    shutil.copy(build.API, tmp_path / "include" / "protnote_hip.h")
    shutil.copy(build.build_lib(verbose=False), pkg / "libprotnote_hip.so")
    return pkg


@pytest.fixture
def scratch(tmp_path, monkeypatch):
    pkg = _clone_sources(tmp_path)
    monkeypatch.setattr(build, "CSRC", str(pkg / "csrc"))
    monkeypatch.setattr(build, "API", str(tmp_path / "include" / "protnote_hip.h"))
    monkeypatch.setattr(build, "LIB", str(pkg / "libprotnote_hip.so"))
    monkeypatch.setattr(build, "OBJ", str(pkg / "csrc" / "_obj"))
    return pkg


def test_built_library_reports_the_hash_of_its_sources():
    from protnote_amd import _lib

    build.build_lib(verbose=False)
    assert not build.stale()
    h = build.csrc_hash()
    assert len(h) == 16 and build.embedded_hash() == h and _lib.build_hash() == h


def test_every_header_is_a_dependency(scratch):
    """Touching ANY csrc/*.hpp (content, not mtime), any translation unit or the API header makes the binary stale."""
    assert not build.stale()
    names = [os.path.basename(f) for f in build.sources()]
    for must in ("gemm_bf16.hpp", "bwd_bf16_dz.hpp", "gemm_engine.hpp", "train_kernels.hpp", "protnote_hip.hip",
                 "metrics.hip", "protnote_hip.h"):
        assert must in names, must
    assert len([n for n in names if n.endswith(".hpp")]) == len(build.headers())
    for f in build.sources():
        orig = open(f, "rb").read()
        try:
            with open(f, "ab") as fh:
                fh.write(b"\n// touched\n")
            assert build.stale(), f
            assert build._unit_stale("build_hash.cpp") or not os.path.exists(build._obj("build_hash.cpp"))
        finally:
            with open(f, "wb") as fh:
                fh.write(orig)
        assert not build.stale(), f


def test_new_header_is_picked_up(scratch):
    (scratch / "csrc" / "brand_new.hpp").write_text("// not yet included anywhere\n")
    assert build.stale()


def test_stale_binary_is_refused_without_a_compiler(scratch, monkeypatch):
    from protnote_amd import _lib

    monkeypatch.setattr(_lib, "LIB_PATH", build.LIB)
    monkeypatch.setattr(build, "have_hipcc", lambda: None)
    monkeypatch.delenv("PN_SKIP_HASH_CHECK", raising=False)
    _lib._ensure_current()  # current binary: accepted
    with open(scratch / "csrc" / "gemm_bf16.hpp", "a") as fh:
        fh.write("\n// edited after the build\n")
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib._ensure_current()
    os.remove(build.LIB)
    with pytest.raises(RuntimeError, match="is missing"):
        _lib._ensure_current()


def test_wrong_hash_in_binary_is_rejected(scratch, monkeypatch):
    from protnote_amd import _lib

    blob = open(build.LIB, "rb").read()
    i = blob.find(build.HASH_MARKER) + len(build.HASH_MARKER)
    with open(build.LIB, "wb") as fh:
        fh.write(blob[:i] + b"0123456789abcdef" + blob[i + 16:])
    assert build.embedded_hash() == "0123456789abcdef" and build.stale()
    monkeypatch.setattr(_lib, "LIB_PATH", build.LIB)
    monkeypatch.setattr(build, "have_hipcc", lambda: None)
    with pytest.raises(RuntimeError, match="0123456789abcdef"):
        _lib._ensure_current()


def test_stale_binary_is_rebuilt_when_hipcc_is_there(scratch, monkeypatch):
    """With a compiler the stale binary is rebuilt, not served (the compile itself is stubbed: it takes a minute)."""
    from protnote_amd import _lib

    calls = []
    monkeypatch.setattr(_lib, "LIB_PATH", build.LIB)
    monkeypatch.setattr(build, "build_lib", lambda verbose=True, force=False: calls.append(1) or build.LIB)
    with open(scratch / "csrc" / "common.hpp", "a") as fh:
        fh.write("\n// edited\n")
    _lib._ensure_current()
    assert calls == [1]


def test_extra_compile_flags_are_part_of_the_hash(scratch, monkeypatch):
    """An A/B build of an experiment macro (PN_EXTRA_HIPCC_FLAGS) is a different binary: the default process rejects it and
    vice versa, so a measurement cannot silently run on the other variant."""
    assert not build.stale()
    h0 = build.csrc_hash()
    monkeypatch.setenv("PN_EXTRA_HIPCC_FLAGS", "-DPN_TN_TASKS_T=1")
    assert build.csrc_hash() != h0 and build.stale()
    monkeypatch.delenv("PN_EXTRA_HIPCC_FLAGS")
    assert build.csrc_hash() == h0 and not build.stale()


def test_objects_newer_than_an_edited_source_are_recompiled(scratch, monkeypatch):
    """file times do not survive the snapshot to the GPU box, so objects can ARRIVE newer than an edited source.
    Each unit's staleness is keyed on content (flags + sha256 of the unit, every header and the API header, kept in
    <obj>.flags): the edited unit is recompiled, not relinked under a fresh build_hash.o.  The compile itself is stubbed."""
    import subprocess
    import time

    os.makedirs(build.OBJ, exist_ok=True)
    for u in build.UNITS:  # a finished build: objects + content stamps
        open(build._obj(u), "wb").write(b"obj")
        open(build._obj(u) + ".flags", "w").write(build._unit_stamp(u))
    assert not any(build._unit_stale(u) for u in build.UNITS)
    with open(scratch / "csrc" / "gemm_bf16.hpp", "a") as fh:
        fh.write("\n// edited after the objects were made\n")
    future = time.time() + 3600
    for u in build.UNITS:  # ... and the objects look NEWER than every source, as after the snapshot
        os.utime(build._obj(u), (future, future))
        os.utime(build._obj(u) + ".flags", (future, future))
    assert all(build._unit_stale(u) for u in build.UNITS)

    compiled = []

    def fake_call(cmd):
        if "-c" in cmd:
            compiled.append(os.path.basename(cmd[cmd.index("-c") + 1]))
            open(cmd[cmd.index("-o") + 1], "wb").write(b"obj")
        else:  # the link: a "binary" that carries the hash the sources have now
            open(cmd[cmd.index("-o") + 1], "wb").write(build.HASH_MARKER + build.csrc_hash().encode())

    monkeypatch.setattr(subprocess, "check_call", fake_call)
    monkeypatch.setattr(build, "have_hipcc", lambda: "/bin/true")
    build.build_lib(verbose=False)
    assert sorted(compiled) == sorted(build.UNITS)
    assert not build.stale() and not any(build._unit_stale(u) for u in build.UNITS)
    # an edit of ONE translation unit recompiles that unit (and the hash carrier), nothing else
    compiled.clear()
    with open(scratch / "csrc" / "metrics.hip", "a") as fh:
        fh.write("\n// edited\n")
    build.build_lib(verbose=False)
    assert sorted(compiled) == ["build_hash.cpp", "metrics.hip"]
