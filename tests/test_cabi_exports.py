"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/protnote_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from protnote_amd.build import build_lib

    return build_lib(verbose=False)


def _declared():
    src = open(os.path.join(ROOT, "include", "protnote_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/protnote_hip.h but not exported"


def test_binding_covers_header(built_lib):
    from protnote_amd import _lib

    assert sorted(_lib.exported_symbols()) == _declared()
    assert _lib.lib().pn_version() >= 1


def test_cpu_tensors_fail_loudly():
    import torch
    from protnote_amd.models.protein_encoders import ProteInfer

    enc = ProteInfer(5, 20, 8, 9, torch.nn.ReLU, 3, 1, 0.5)
    for p in enc.parameters():
        p.requires_grad = False
    with pytest.raises(RuntimeError, match="HIP device"):
        enc.get_embeddings(torch.zeros(1, 20, 16), torch.tensor([16]))


def test_struct_sizes_match_c(built_lib, tmp_path):
    """ctypes mirrors of the descriptor structs must have the C compiler's layout."""
    import subprocess
    from protnote_amd import _lib

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "protnote_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   "sizeof(pn_bn),sizeof(pn_res_block),sizeof(pn_encoder),sizeof(pn_mlp),sizeof(pn_pairhead));}")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(t) for t in (_lib.pn_bn, _lib.pn_res_block, _lib.pn_encoder, _lib.pn_mlp,
                                       _lib.pn_pairhead)]
    assert got == want


def test_error_convention(built_lib):
    """Entry points return non-zero and leave a message in pn_last_error() on invalid arguments (checked before any
    launch, so this runs without a GPU)."""
    from protnote_amd import _lib

    lib = _lib.lib()
    rc = lib.pn_gemm_nt(None, 6, None, 6, None, 8, 4, 8, 6, None, None, None, None, None, 0, None, 0, None)  # K % 4 != 0
    assert rc != 0 and b"multiples of 4" in lib.pn_last_error()
    rc = lib.pn_ensemble_logit(None, 4, 10, 3, 0, None, None)  # NL not divisible by ndesc
    assert rc != 0 and b"divisible" in lib.pn_last_error()
    with pytest.raises(RuntimeError, match="libprotnote_hip"):
        _lib.check(rc)
    # descriptor fields are validated before anything is launched: an unknown forward_math / math_mode is an error
    hd = _lib.pn_pairhead()
    hd.nlayers, hd.h, hd.d, hd.math_mode, hd.forward_math = 3, 256, 64, 1, 7
    rc = lib.pn_pairhead_fwd_eval(ctypes.byref(hd), None, None, 4, 4, None, 0, None, 0, None)
    assert rc != 0 and b"forward_math" in lib.pn_last_error()
    hd.forward_math, hd.math_mode = 2, 9
    rc = lib.pn_pairhead_fwd_eval(ctypes.byref(hd), None, None, 4, 4, None, 0, None, 0, None)
    assert rc != 0 and b"math_mode" in lib.pn_last_error()


def test_grad_struct_sizes_match_c(built_lib, tmp_path):
    import subprocess
    from protnote_amd import _lib

    src = tmp_path / "sz2.c"
    src.write_text('#include <stdio.h>\n#include "protnote_hip.h"\nint main(){printf("%zu %zu\\n",'
                   "sizeof(pn_mlp_grads),sizeof(pn_pairhead_grads));}")
    exe = tmp_path / "sz2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(_lib.pn_mlp_grads), ctypes.sizeof(_lib.pn_pairhead_grads)]
