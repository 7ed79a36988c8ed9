"""ctypes binding of libprotnote_hip.so (the C ABI in include/protnote_hip.h).

The library is mandatory: importing a kernel entry point without the built .so raises - there is no
CPU or eager fallback anywhere in protnote_amd."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprotnote_hip.so")

PN_MAX_BLOCKS = 16
PN_MAX_LAYERS = 8
PN_ADAM_WS_BYTES = 32768
fp = C.POINTER(C.c_float)


class pn_bn(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p)]


class pn_res_block(C.Structure):
    _fields_ = [("bn1", pn_bn), ("conv_a_w", C.c_void_p), ("conv_a_b", C.c_void_p), ("bn2", pn_bn),
                ("conv_b_w", C.c_void_p), ("conv_b_b", C.c_void_p)]


class pn_encoder(C.Structure):
    _fields_ = [("Cin", C.c_int), ("C", C.c_int), ("Cb", C.c_int), ("ksize", C.c_int), ("nblocks", C.c_int),
                ("dil_base", C.c_int), ("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p),
                ("blk", pn_res_block * PN_MAX_BLOCKS), ("bn_use_running", C.c_int), ("math_mode", C.c_int)]


class pn_mlp(C.Structure):
    _fields_ = [("nlayers", C.c_int), ("dims", C.c_int * (PN_MAX_LAYERS + 1)),
                ("w", C.c_void_p * PN_MAX_LAYERS), ("bias", C.c_void_p * PN_MAX_LAYERS),
                ("bn", pn_bn * PN_MAX_LAYERS), ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
                ("dropout_p", C.c_float), ("dropout_seed", C.c_uint), ("dropout_stream", C.c_int),
                ("bn_use_running", C.c_int), ("math_mode", C.c_int)]


class pn_pairhead(C.Structure):
    _fields_ = [("d", C.c_int), ("in_dim", C.c_int), ("fusion", C.c_int), ("nlayers", C.c_int), ("h", C.c_int),
                ("w", C.c_void_p * PN_MAX_LAYERS), ("bias", C.c_void_p * PN_MAX_LAYERS),
                ("bn", pn_bn * PN_MAX_LAYERS), ("w_out", C.c_void_p), ("b_out", C.c_void_p),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint),
                ("bn_use_running", C.c_int), ("math_mode", C.c_int), ("backward_math", C.c_int),
                ("forward_math", C.c_int)]


class pn_res_block_grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("bn1_w", "bn1_b", "conv_a_w", "conv_a_b", "bn2_w", "bn2_b", "conv_b_w",
                                          "conv_b_b")]


class pn_encoder_grads(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("blk", pn_res_block_grads * PN_MAX_BLOCKS)]


class pn_mlp_grads(C.Structure):
    _fields_ = [("dw", C.c_void_p * PN_MAX_LAYERS), ("dgamma", C.c_void_p * PN_MAX_LAYERS),
                ("dbeta", C.c_void_p * PN_MAX_LAYERS)]


class pn_pairhead_grads(C.Structure):
    _fields_ = [("dw", C.c_void_p * PN_MAX_LAYERS), ("dgamma", C.c_void_p * PN_MAX_LAYERS),
                ("dbeta", C.c_void_p * PN_MAX_LAYERS), ("dw_out", C.c_void_p), ("db_out", C.c_void_p)]


_lib = None

_SIGS = {
    "pn_last_error": (C.c_char_p, []),
    "pn_version": (C.c_int, []),
    "pn_build_hash": (C.c_char_p, []),
    "pn_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pn_encoder_ws_bytes": (C.c_size_t, [C.POINTER(pn_encoder), C.c_int, C.c_int]),
    "pn_encoder_fwd": (C.c_int, [C.POINTER(pn_encoder), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                 C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_masked_conv1d_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pn_masked_conv1d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_residual_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pn_residual_fwd": (C.c_int, [C.POINTER(pn_res_block), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_mlp_rows_ws_bytes": (C.c_size_t, [C.POINTER(pn_mlp), C.c_int]),
    "pn_mlp_rows_fwd_eval": (C.c_int, [C.POINTER(pn_mlp), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "pn_pairhead_eval_ws_bytes": (C.c_size_t, [C.POINTER(pn_pairhead), C.c_int, C.c_int, C.c_int]),
    "pn_pairhead_fwd_eval": (C.c_int, [C.POINTER(pn_pairhead), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_pairhead_hidden_ws_bytes": (C.c_size_t, [C.POINTER(pn_pairhead), C.c_int, C.c_int]),
    "pn_pairhead_fwd_eval_hidden": (C.c_int, [C.POINTER(pn_pairhead), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_additive_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "pn_additive_attention_bwd_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pn_additive_attention_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_ensemble_logit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pn_ensemble_logit_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pn_label_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_long, C.c_void_p]),
    "pn_label_noise_seeded": (C.c_int, [C.c_void_p, C.c_uint, C.c_float, C.c_void_p, C.c_long, C.c_int, C.c_void_p]),
    "pn_uniform": (C.c_int, [C.c_uint, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "pn_encoder_fwd_ids": (C.c_int, [C.POINTER(pn_encoder), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_similarity_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pn_similarity_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_encoder_train_save_bytes": (C.c_size_t, [C.POINTER(pn_encoder), C.c_int, C.c_int]),
    "pn_encoder_bwd_ws_bytes": (C.c_size_t, [C.POINTER(pn_encoder), C.c_int, C.c_int]),
    "pn_encoder_fwd_train": (C.c_int, [C.POINTER(pn_encoder), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_encoder_bwd": (C.c_int, [C.POINTER(pn_encoder), C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.POINTER(pn_encoder_grads), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "pn_mlp_rows_train_save_bytes": (C.c_size_t, [C.POINTER(pn_mlp), C.c_int]),
    "pn_mlp_rows_train_ws_bytes": (C.c_size_t, [C.POINTER(pn_mlp), C.c_int]),
    "pn_mlp_rows_fwd_train": (C.c_int, [C.POINTER(pn_mlp), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_mlp_rows_bwd": (C.c_int, [C.POINTER(pn_mlp), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.POINTER(pn_mlp_grads), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "pn_pairhead_train_save_bytes": (C.c_size_t, [C.POINTER(pn_pairhead), C.c_int, C.c_int, C.c_int]),
    "pn_pairhead_train_ws_bytes": (C.c_size_t, [C.POINTER(pn_pairhead), C.c_int, C.c_int]),
    "pn_pairhead_fwd_train": (C.c_int, [C.POINTER(pn_pairhead), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "pn_pairhead_bwd": (C.c_int, [C.POINTER(pn_pairhead), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.POINTER(pn_pairhead_grads), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_pairhead_train_hidden": (C.c_int, [C.POINTER(pn_pairhead), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p]),
    "pn_similarity_train_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pn_similarity_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_loss_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    "pn_loss_fwd_bwd_t": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "pn_tp_fn_fp_t": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "pn_supcon_ws_bytes": (C.c_size_t, [C.c_int]),
    "pn_supcon_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_tp_fn_fp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "pn_clip_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "pn_clip_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_transpose": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p]),
    "pn_gemm_tn": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int,
                             C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_onehot_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "pn_ap_append": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    "pn_ap_ws_bytes": (C.c_size_t, [C.c_int, C.c_longlong, C.c_longlong, C.c_int]),
    "pn_ap_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_ap_partial_ws_bytes": (C.c_size_t, [C.c_longlong]),
    "pn_ap_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_binned_hist_update": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pn_binned_auprc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p]),
    "pn_dropout_mask": (C.c_int, [C.c_uint, C.c_int, C.c_float, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "pn_set_math_mode": (C.c_int, [C.c_int]),
    "pn_set_mlp_materialize": (C.c_int, [C.c_int]),
    "pn_set_sync_bn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]),
    "pn_get_math_mode": (C.c_int, []),
    "pn_set_backward_math": (C.c_int, [C.c_int]),
    "pn_get_backward_math": (C.c_int, []),
    "pn_set_forward_math": (C.c_int, [C.c_int]),
    "pn_get_forward_math": (C.c_int, []),
    "pn_set_fwd_staged": (C.c_int, [C.c_int]),
    "pn_set_bf16_mfma16": (C.c_int, [C.c_int]),
    "pn_set_bwd_deep": (C.c_int, [C.c_int]),
    "pn_set_f32_dma": (C.c_int, [C.c_int]),
    "pn_set_encoder_f64": (C.c_int, [C.c_int]),
    "pn_set_conv1_gather": (C.c_int, [C.c_int]),
    "pn_set_b3_dma": (C.c_int, [C.c_int]),
    "pn_prof_begin": (C.c_int, []),
    "pn_prof_end": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]),
    "pn_gemm_nt": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int,
                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_void_p, C.c_size_t, C.c_void_p]),
    "pn_gemm_nt_stats_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pn_loss_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
}


def exported_symbols():
    return sorted(_SIGS)


def _ensure_current():
    """The .so must exist AND be built from the sources beside it.  With csrc/ present the content hash stamped into the
    binary (pn_build_hash) is compared with build.csrc_hash(): a missing or stale binary is rebuilt when hipcc is there
    and refused otherwise - never served silently (a stale .so once produced a null experiment)."""
    from . import build

    have_src = os.path.isdir(build.CSRC) and os.path.exists(os.path.join(build.CSRC, "protnote_hip.hip"))
    if os.path.exists(LIB_PATH) and (not have_src or os.environ.get("PN_SKIP_HASH_CHECK") == "1"):
        return
    if os.path.exists(LIB_PATH) and not build.stale():
        return
    why = "is missing" if not os.path.exists(LIB_PATH) else (
        f"was built from other sources (binary {build.embedded_hash()}, csrc/ {build.csrc_hash()})")
    if not build.have_hipcc():
        raise RuntimeError(f"{LIB_PATH} {why} and hipcc is not available to rebuild it; build it with "
                           "`python -m protnote_amd.build` (protnote_amd has no CPU/eager fallback)")
    try:  # in-tree build with hipcc; never a CPU/eager fallback
        build.build_lib(verbose=False)
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError(f"{LIB_PATH} {why} and could not be rebuilt ({exc}); build it with "
                           "`python -m protnote_amd.build` (protnote_amd has no CPU/eager fallback)") from exc


def build_hash() -> str:
    """Source hash compiled into the loaded library."""
    return lib().pn_build_hash().decode()


def lib():
    """Load (once) and return the C-ABI library; raises if it is missing or stale and cannot be rebuilt."""
    global _lib
    if _lib is None:
        _ensure_current()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
        mode = os.environ.get("PN_MATH_MODE")
        if mode:
            set_math_mode(mode)
        mode = os.environ.get("PN_BACKWARD_MATH")
        if mode:
            set_backward_math(mode)
        mode = os.environ.get("PN_FORWARD_MATH")
        if mode:
            set_forward_math(mode)
    return _lib


_MATH_MODES = {"f32": 0, "0": 0, "bf16x3": 1, "1": 1}


def set_math_mode(mode) -> None:
    """Arithmetic of the pair-grid GEMMs: "f32" (default, exact f32 MFMA) or "bf16x3" (split-bf16 products with f32
    accumulation: ~1e-5 relative error per product, several times faster).  Also settable with PN_MATH_MODE."""
    key = str(mode).lower()
    if key not in _MATH_MODES:
        raise ValueError(f"math mode must be 'f32' or 'bf16x3', got {mode!r}")
    check(lib().pn_set_math_mode(_MATH_MODES[key]))


def get_math_mode() -> str:
    return "bf16x3" if lib().pn_get_math_mode() == 1 else "f32"


def math_field(mode=None) -> int:
    """Value of a descriptor's `math_mode` field (include/protnote_hip.h: 1 = f32, 2 = bf16x3).  The mode travels WITH the
    call: `mode` None reads the process default (set_math_mode / PN_MATH_MODE) now, at descriptor-build time, so the C side
    never consults its global for a call made through this binding - two host threads driving two models can run different
    modes (set `model.math_mode`), and flipping the default on one thread cannot change a call another thread is making."""
    if mode is None:
        return 1 + int(lib().pn_get_math_mode())
    key = str(mode).lower()
    if key not in _MATH_MODES:
        raise ValueError(f"math mode must be 'f32' or 'bf16x3', got {mode!r}")
    return 1 + _MATH_MODES[key]


_BWD_MODES = {"same": 0, "0": 0, "bf16": 1, "1": 1}


def set_backward_math(mode) -> None:
    """Arithmetic of the backward pair-grid GEMMs of the output MLP's hidden layers: "same" (default: the forward's mode)
    or "bf16" (one bf16 product, f32 accumulation - the class of the reference's autocast backward).  The forward and
    the logits are not affected.  Also settable with PN_BACKWARD_MATH."""
    key = str(mode).lower()
    if key not in _BWD_MODES:
        raise ValueError(f"backward math must be 'same' or 'bf16', got {mode!r}")
    check(lib().pn_set_backward_math(_BWD_MODES[key]))


def get_backward_math() -> str:
    return "bf16" if lib().pn_get_backward_math() == 1 else "same"


def backward_math_field(mode=None) -> int:
    """Value of pn_pairhead.backward_math (1 = as the forward, 2 = bf16); None = the process default, read now."""
    if mode is None:
        return 1 + int(lib().pn_get_backward_math())
    key = str(mode).lower()
    if key not in _BWD_MODES:
        raise ValueError(f"backward math must be 'same' or 'bf16', got {mode!r}")
    return 1 + _BWD_MODES[key]


def set_forward_math(mode) -> None:
    """Arithmetic of the FORWARD pair-grid GEMMs of the output MLP's hidden layers: "same" (default: math_mode's kernels)
    or "bf16" (one bf16 product, f32 accumulation - the class of the reference's autocast forward,
    ProtNoteTrainer.py:287,728-729).  Opt-in: logits move by ~1e-2 at O(1) scale (weight rounding), so this mode is held
    to torch's own autocast(bfloat16) run of the oracle, not to the 1e-3 bound of "f32" / "bf16x3".  Stored activations,
    BatchNorm statistics and the loss stay f32.  Also settable with PN_FORWARD_MATH."""
    key = str(mode).lower()
    if key not in _BWD_MODES:
        raise ValueError(f"forward math must be 'same' or 'bf16', got {mode!r}")
    check(lib().pn_set_forward_math(_BWD_MODES[key]))


def get_forward_math() -> str:
    return "bf16" if lib().pn_get_forward_math() == 1 else "same"


def forward_math_field(mode=None) -> int:
    """Value of pn_pairhead.forward_math (1 = as math_mode, 2 = bf16); None = the process default, read now."""
    if mode is None:
        return 1 + int(lib().pn_get_forward_math())
    key = str(mode).lower()
    if key not in _BWD_MODES:
        raise ValueError(f"forward math must be 'same' or 'bf16', got {mode!r}")
    return 1 + _BWD_MODES[key]


PN_LABEL_F32, PN_LABEL_I64, PN_LABEL_U8 = 0, 1, 2


def typed_targets(target):
    """(tensor to keep alive, PN_LABEL_* kind) for a multihot target tensor: int64 (the reference collator's dtype) and uint8 /
    bool (1 B per pair) go to the kernels as they are, anything else as float32."""
    if target.dtype == torch.int64:
        return target.contiguous(), PN_LABEL_I64
    if target.dtype in (torch.uint8, torch.bool):
        t = target.contiguous()
        return (t.view(torch.uint8) if t.dtype == torch.bool else t), PN_LABEL_U8
    return target.detach().float().contiguous(), PN_LABEL_F32


def check(rc: int):
    if rc != 0:
        raise RuntimeError("libprotnote_hip: " + lib().pn_last_error().decode())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("protnote_amd runs on an MI355X HIP device only: got a CPU tensor "
                               "(there is no CPU fallback; move the model and inputs to cuda)")


_ws_cache = {}


def workspace(nbytes: int, device, tag: str = "") -> torch.Tensor:
    """Grow-only scratch buffer (uint8) per (device, stream, tag): calls queued on one stream reuse one buffer in stream
    order; calls on ANOTHER stream (a second model training concurrently, a second host thread) get their own, so the
    C ABI's "thread-safe per stream" holds for the workspaces this binding hands it (pacing counters included)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _ws_cache.pop(key, None)
        if nbytes > (8 << 30):  # same reason as train_path._save_buf: do not keep the outgrown block cached
            torch.cuda.empty_cache()
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def prof_begin():
    check(lib().pn_prof_begin())


def prof_end():
    """-> {kind: (launches, total_ms, total_flops)} (synchronises the device first)."""
    torch.cuda.synchronize()
    n = 128
    kinds, counts = (C.c_int * n)(), (C.c_long * n)()
    ms, fl = (C.c_double * n)(), (C.c_double * n)()
    got = lib().pn_prof_end(n, kinds, counts, ms, fl)
    return {int(kinds[i]): (int(counts[i]), float(ms[i]), float(fl[i])) for i in range(got)}
