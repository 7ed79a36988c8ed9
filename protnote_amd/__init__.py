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
