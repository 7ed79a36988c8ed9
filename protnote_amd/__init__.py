"""protnote_amd - MI355X-native implementation of ProtNote's forward/training hot path.

Module layout mirrors the reference package so `protnote.models.ProtNote.ProtNote`,
`protnote.models.protein_encoders.ProteInfer` and `protnote.utils.losses.get_loss` have drop-in twins
under `protnote_amd.`.  All arithmetic runs in hand-written HIP kernels reached
through the C ABI in include/protnote_hip.h; PyTorch only owns device memory, streams and
torch.distributed."""
__version__ = "0.1.0"


def free_workspaces():
    """Drop the cached device scratch buffers (grow-only per-device workspaces of the C-ABI calls).  The
    training-time activation store of a model lives on the model (`model._pn_train_save`) and goes with it."""
    from . import _lib

    _lib._ws_cache.clear()


def set_math_mode(mode) -> None:
    """"f32" (default) or "bf16x3" for the pair-grid GEMMs; see include/protnote_hip.h pn_set_math_mode."""
    from . import _lib

    _lib.set_math_mode(mode)


def get_math_mode() -> str:
    from . import _lib

    return _lib.get_math_mode()


def set_backward_math(mode) -> None:
    """"same" (default) or "bf16" for the backward pair-grid GEMMs; see include/protnote_hip.h pn_set_backward_math."""
    from . import _lib

    _lib.set_backward_math(mode)


def get_backward_math() -> str:
    from . import _lib

    return _lib.get_backward_math()


def set_forward_math(mode) -> None:
    """"same" (default) or "bf16" for the forward pair-grid GEMMs of the output MLP's hidden layers (opt-in AMP class); see
    include/protnote_hip.h pn_set_forward_math."""
    from . import _lib

    _lib.set_forward_math(mode)


def get_forward_math() -> str:
    from . import _lib

    return _lib.get_forward_math()


# reference module path -> twin module; a reference module without an entry here is outside the
# hot path and is NOT aliased
_PROTNOTE_ALIASES = {
    "protnote": "protnote_amd",
    "protnote.models": "protnote_amd.models",
    "protnote.models.ProtNote": "protnote_amd.models.ProtNote",
    "protnote.models.protein_encoders": "protnote_amd.models.protein_encoders",
    "protnote.models.ProtNoteTrainer": "protnote_amd.models.ProtNoteTrainer",
    "protnote.utils": "protnote_amd.utils",
    "protnote.utils.losses": "protnote_amd.utils.losses",
    "protnote.utils.models": "protnote_amd.utils.models",
    "protnote.utils.configs": "protnote_amd.utils.configs",
    "protnote.utils.proteinfer": "protnote_amd.utils.proteinfer",
    "protnote.utils.evaluation": "protnote_amd.utils.evaluation",
    "protnote.data": "protnote_amd.data",
    "protnote.data.collators": "protnote_amd.data.collators",
    "protnote.data.samplers": "protnote_amd.data.samplers",
}


def install_as_protnote(force: bool = False):
    """Make `from protnote.models.ProtNote import ProtNote`, `from protnote.models.protein_encoders import ProteInfer`,
    `from protnote.utils.losses import get_loss` (the imports of reference bin/main.py:9-11) and the other hot-path module
    paths resolve to the protnote_amd twins: the `sys.modules` aliasing of INTEGRATION.md section 1, as one call made before
    the caller's own imports.  Refuses to shadow a real `protnote` package that is already imported (or importable) unless
    `force=True`.  Returns the list of aliased module names."""
    import importlib
    import importlib.util
    import sys

    if not force:
        mod = sys.modules.get("protnote")
        if mod is not None and getattr(mod, "__name__", "") != "protnote_amd":
            raise RuntimeError("a different `protnote` package is already imported; call install_as_protnote(force=True) "
                               "before importing it, or remove it from the path")
        if mod is None and importlib.util.find_spec("protnote") is not None:
            raise RuntimeError("a real `protnote` package is importable from sys.path; install_as_protnote(force=True) "
                               "shadows it for this process")
    done = []
    for ref, twin in _PROTNOTE_ALIASES.items():
        sys.modules[ref] = importlib.import_module(twin)
        done.append(ref)
    return done
