"""Training-mode forward/backward of the ProtNote heads on MI355X (reference ProtNote.forward under
model.train(), ProtNote.py:168-334, driven by ProtNoteTrainer.py:728-738).

The autograd boundary is one torch.autograd.Function around the three trainable stacks (W_p, W_l,
output_layer): forward and backward are sequences of C-ABI calls (pn_mlp_rows_fwd_train,
pn_pairhead_fwd_train, pn_pairhead_bwd, pn_mlp_rows_bwd); torch only routes the returned gradient tensors.
The frozen encoder runs under no_grad with train-mode BatchNorm exactly like the reference."""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from .. import _lib as L


def _flat_modules(seq):
    out = []

    def walk(m):
        if isinstance(m, nn.Sequential):
            for c in m:
                walk(c)
        else:
            out.append(m)

    walk(seq)
    return out


def _stack_params(layers):
    """Canonical parameter order of a [(Linear, BN|None)] stack: lin.weight, [lin.bias], [bn.weight, bn.bias]."""
    ps = []
    for lin, bn in layers:
        ps.append(lin.weight)
        if lin.bias is not None:
            ps.append(lin.bias)
        if bn is not None:
            ps += [bn.weight, bn.bias]
    return ps


def _backward_pending(model) -> bool:
    """True while the graph of the model's last differentiable train-path forward is still alive (its backward has not
    run and its context has not been collected): the one activation store is then spoken for."""
    ref = model.__dict__.get("_pn_train_pending")
    return ref is not None and ref() is not None


def _save_buf(model, tag, nbytes, device, temporary=False, private=None):
    if private is not None:  # a store owned by ONE forward's autograd context (see _HeadsTrainFn.forward)
        buf = private.get(tag)
        if buf is None:  # carved from the block _try_private_store allocated (and proved allocatable) up front
            start = (private.used + 255) // 256 * 256
            if start + int(nbytes) > private.block.numel():
                raise RuntimeError("private activation store too small for this forward (internal sizing error)")
            buf = private[tag] = private.block[start:start + int(nbytes)]
            private.used = start + int(nbytes)
        return buf
    if temporary:  # a no_grad forward while another forward's backward is pending: do not touch the shared store
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    cache = model.__dict__.setdefault("_pn_train_save", {})
    buf = cache.get(tag)
    if buf is None or buf.numel() < nbytes or buf.device != device:
        cache.pop(tag, None)
        buf = None
        if nbytes > (8 << 30):  # growing a multi-GB store: give the old block back to the driver first, torch's
            torch.cuda.empty_cache()  # caching allocator would otherwise keep it next to the new, larger one
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        cache[tag] = buf
    return buf


def _try_private_store(model, P_f, L_f):
    """Room for a second set of saved activations?  Returns an (empty) private store when the three save buffers of this
    forward fit in what the device has free (the driver's free memory plus torch's cached-but-unused blocks) with 10 %
    to spare, else None."""
    lib = L.lib()
    dev = P_f.device
    B, NL = P_f.shape[0], L_f.shape[0]
    mp, _ = model._mlp_desc(model.W_p, None, 100)
    ml, _ = model._mlp_desc(model.W_l, None, 200)
    need = lib.pn_mlp_rows_train_save_bytes(C.byref(mp), B) + lib.pn_mlp_rows_train_save_bytes(C.byref(ml), NL)
    if model.feature_fusion != "similarity":
        hd, _ = model._pair_desc(None)
        need += lib.pn_pairhead_train_save_bytes(C.byref(hd), B, NL, model._train_chunk(B, NL))
    free, _total = torch.cuda.mem_get_info(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    # every private store is memory a caller holds for as long as it keeps grad-enabled logits alive (collecting outputs
    # in a list without calling backward): cap how many can exist, beyond that the shared store is reused
    live = model.__dict__.setdefault("_pn_private_stores", weakref.WeakSet())
    if len(live) >= MAX_PRIVATE_STORES or need * 1.1 >= free + cached:
        return None
    try:  # free + cached can be fragmented: the allocation itself decides, and failing it means "use the shared store"
        store = _PrivateStore(torch.empty(int(need) + 4 * 256, dtype=torch.uint8, device=dev))
    except torch.cuda.OutOfMemoryError:
        return None
    live.add(store)
    return store


MAX_PRIVATE_STORES = 2


class _PrivateStore(dict):
    """Save buffers of one differentiable forward that overlaps an earlier one (released with its autograd context)."""
    __slots__ = ("block", "used", "__weakref__")

    def __init__(self, block):
        super().__init__()
        self.block, self.used = block, 0

    __hash__ = object.__hash__  # dict subclasses are unhashable by default; the WeakSet of live stores needs identity
    __eq__ = object.__eq__


class _HeadsTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, P_f, L_f, opts, *params):
        """`opts` (a plain dict owned by the caller of this ONE forward - nothing is parked on the model, so concurrent forwards
        on one model cannot steal each other's flags): "want_graph" = autograd is recording (this function itself
        runs under no_grad), "want_embeddings" = save_embeddings=True, "embeddings" = where they are returned."""
        lib = L.lib()
        dev = P_f.device
        st = L.stream_ptr()
        ctx.model = model
        ctx.param_list = params
        want_graph = bool(opts.get("want_graph", True))
        # model.eval() + autograd (reference ProtNote.forward has no mode restriction): BatchNorm normalises with its
        # running statistics and updates nothing (pn_mlp.bn_use_running)
        bn_running = ctx.bn_running = 0 if model.training else 1
        # The saved activations of the usual one-forward-one-backward loop live in ONE grow-only buffer per model (2 x 101
        # GB at the bench config).  A differentiable forward that starts while an earlier forward's backward is still
        # pending gets a store of its OWN, held by its autograd context and released with it (torch semantics: both
        # backwards work, in any order) - if the device has the memory; if it has not (two bench-size stores cannot
        # exist), the new forward takes the shared store and the earlier backward raises.  The stamp below is what
        # that backward checks.  A forward under torch.no_grad() (train-mode BatchNorm still advances its buffers,
        # SURVEY 3.4-1) has no backward: it reuses the shared store when it is free and takes a temporary one otherwise.
        temporary = False
        ctx.own_save = None
        B, NL = P_f.shape[0], L_f.shape[0]
        if want_graph and _backward_pending(model):
            ctx.own_save = _try_private_store(model, P_f, L_f)
        if ctx.own_save is not None:
            ctx.generation = None
        elif want_graph:
            ctx.generation = model.__dict__["_pn_train_generation"] = model.__dict__.get("_pn_train_generation", 0) + 1
            try:
                model.__dict__["_pn_train_pending"] = weakref.ref(ctx)
            except TypeError:  # context objects without weak-reference support: assume pending until a backward runs
                model.__dict__["_pn_train_pending"] = lambda: True
        else:
            ctx.generation = None
            temporary = _backward_pending(model)
        own = ctx.own_save
        # OUTPUT_MLP_DROPOUT: one fresh seed per forward (host RNG: follows torch.manual_seed, no device sync); the
        # backward regenerates the same masks from it.  Dropout layers are the identity in eval mode.
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (model.mlp_dropout > 0 and model.training) else None
        ctx.drop_seed = seed
        mp, lp = model._mlp_desc(model.W_p, seed, 100)
        ml, ll = model._mlp_desc(model.W_l, seed, 200)
        mp.bn_use_running = ml.bn_use_running = bn_running
        # the arithmetic of this step travels with it: the backward re-builds its descriptors with the forward's modes even
        # if the model's (or the process default's) mode is changed in between
        ctx.math_field = mp.math_mode
        ctx.bwd_field = L.backward_math_field(model.backward_math)
        ctx.P_f, ctx.L_f = P_f, L_f

        def mlp_fwd(m, x, tag):
            rows = x.shape[0]
            save = _save_buf(model, tag, lib.pn_mlp_rows_train_save_bytes(C.byref(m), rows), dev, temporary, own)
            ws = L.workspace(lib.pn_mlp_rows_train_ws_bytes(C.byref(m), rows), dev, "train")
            y = torch.empty(rows, m.dims[m.nlayers], dtype=torch.float32, device=dev)
            L.check(lib.pn_mlp_rows_fwd_train(C.byref(m), L.ptr(x), x.shape[1], rows, L.ptr(y), L.ptr(save),
                                              save.numel(), L.ptr(ws), ws.numel(), st))
            return y

        P_e = mlp_fwd(mp, P_f, "W_p")
        L_e = mlp_fwd(ml, L_f, "W_l")
        ctx.P_e, ctx.L_e = P_e, L_e
        # BatchNorm bookkeeping of all three stacks in ONE multi-tensor add (was one tiny launch per BatchNorm)
        tracked = [bn.num_batches_tracked for _, bn in lp + ll if bn is not None] if model.training else []

        if model.feature_fusion == "similarity":
            if tracked:  # PROJECTION_HEAD_NUM_LAYERS: 1 -> no BatchNorm in W_p / W_l
                torch._foreach_add_(tracked, 1)
            return model._similarity(P_e, L_e)
        hd, hl = model._pair_desc(seed)
        hd.bn_use_running = bn_running
        chunk = model._train_chunk(B, NL)
        ctx.chunk = chunk
        save = _save_buf(model, "pair", lib.pn_pairhead_train_save_bytes(C.byref(hd), B, NL, chunk), dev, temporary, own)
        ws = L.workspace(lib.pn_pairhead_train_ws_bytes(C.byref(hd), B, NL), dev, "train")
        pairs = torch.empty(NL * B, dtype=torch.float32, device=dev)
        L.check(lib.pn_pairhead_fwd_train(C.byref(hd), L.ptr(P_e), L.ptr(L_e), B, NL, L.ptr(pairs), chunk,
                                          L.ptr(save), save.numel(), L.ptr(ws), ws.numel(), st))
        if opts.get("want_embeddings", False):
            # save_embeddings=True (reference ProtNote.py:292-302,324-332): the penultimate output-MLP activations are read
            # back from the store this forward just filled; rows in the reference's protein-major order i * N_L + j
            hidden = torch.empty(NL, B, hd.h, dtype=torch.float32, device=dev)
            L.check(lib.pn_pairhead_train_hidden(C.byref(hd), B, NL, chunk, L.ptr(save), save.numel(), L.ptr(hidden), st))
            opts["embeddings"] = {
                "output_layer_embeddings": hidden.permute(1, 0, 2).reshape(B * NL, hd.h).cpu(),
                "joint_embeddings": model._joint_embeddings_cpu(P_e, L_e)}
            del hidden
        if model.training:
            tracked += [bn.num_batches_tracked for _, bn in hl[:-1] if bn is not None]
        if tracked:
            torch._foreach_add_(tracked, 1)
        logits = torch.empty(B, NL, dtype=torch.float32, device=dev)
        L.check(lib.pn_transpose(L.ptr(pairs), B, NL, B, L.ptr(logits), NL, st))
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if model is None:
            raise RuntimeError("protnote_amd: backward called twice on one train-mode forward (the saved activations "
                               "are consumed in place; retain_graph is not supported)")
        if ctx.own_save is None:
            if model.__dict__.get("_pn_train_generation") != ctx.generation:
                raise RuntimeError("protnote_amd: another differentiable forward ran on this model before this backward and "
                                   "the device had no room for a second activation store, so the saved activations were "
                                   "overwritten.  Call backward() after each forward (gradient accumulation does exactly "
                                   "that), or run the extra forwards under torch.no_grad()")
            model.__dict__["_pn_train_pending"] = None
        lib = L.lib()
        st = L.stream_ptr()
        P_f, L_f, P_e, L_e = ctx.P_f, ctx.L_f, ctx.P_e, ctx.L_e
        dev = P_f.device
        B, NL = P_f.shape[0], L_f.shape[0]
        dlogits = dlogits.contiguous().float()

        grads = {}
        # parameters with requires_grad False (TRAIN_PROJECTION_HEAD: False -> output_layer.*, ProtNoteTrainer.py:221-222,
        # or frozen by hand) get a NULL destination: the C side then skips their gradient GEMMs / reductions entirely
        wanted = {id(p) for p, need in zip(ctx.param_list, ctx.needs_input_grad[4:]) if need}

        def gbuf(p):
            if id(p) not in wanted:
                return None
            g = torch.empty_like(p, memory_format=torch.contiguous_format)
            grads[id(p)] = g
            return g.data_ptr()

        if model.feature_fusion == "similarity":
            d = P_e.shape[1]
            dP_e = torch.empty_like(P_e)
            dL_e = torch.empty_like(L_e)
            ws = L.workspace(lib.pn_similarity_train_ws_bytes(B, NL, d), dev, "train")
            L.check(lib.pn_similarity_bwd(L.ptr(P_e), L.ptr(L_e), B, NL, d, float(model.temperature), L.ptr(dlogits),
                                          L.ptr(dP_e), L.ptr(dL_e), L.ptr(ws), ws.numel(), st))
            return _HeadsTrainFn._finish(ctx, model, lib, st, dev, P_f, L_f, dP_e, dL_e, grads, gbuf)

        dl_pairs = torch.empty(NL * B, dtype=torch.float32, device=dev)
        L.check(lib.pn_transpose(L.ptr(dlogits), NL, B, NL, L.ptr(dl_pairs), B, st))

        # ---- pair head ----
        hd, hl = model._pair_desc(ctx.drop_seed)
        hd.bn_use_running = ctx.bn_running
        hd.math_mode, hd.backward_math = ctx.math_field, ctx.bwd_field
        hidden, out = hl[:-1], hl[-1][0]
        gr = L.pn_pairhead_grads()
        for i, (lin, bn) in enumerate(hidden):
            gr.dw[i] = gbuf(lin.weight)
            if bn is not None:
                gr.dgamma[i] = gbuf(bn.weight)
                gr.dbeta[i] = gbuf(bn.bias)
            elif lin.bias is not None:  # OUTPUT_MLP_BATCHNORM: False - the slot carries the Linear bias gradient
                gr.dbeta[i] = gbuf(lin.bias)
        gr.dw_out = gbuf(out.weight)
        gr.db_out = gbuf(out.bias)
        dP_e = torch.empty_like(P_e)
        dL_e = torch.empty_like(L_e)
        save = _save_buf(model, "pair", 0, dev, private=ctx.own_save)
        ws = L.workspace(lib.pn_pairhead_train_ws_bytes(C.byref(hd), B, NL), dev, "train")
        L.check(lib.pn_pairhead_bwd(C.byref(hd), L.ptr(P_e), L.ptr(L_e), B, NL, L.ptr(dl_pairs), C.byref(gr),
                                    L.ptr(dP_e), L.ptr(dL_e), ctx.chunk, L.ptr(save), save.numel(), L.ptr(ws),
                                    ws.numel(), st))

        return _HeadsTrainFn._finish(ctx, model, lib, st, dev, P_f, L_f, dP_e, dL_e, grads, gbuf)

    @staticmethod
    def _finish(ctx, model, lib, st, dev, P_f, L_f, dP_e, dL_e, grads, gbuf):
        # ---- projection heads ----
        def mlp_bwd(seq, x, dy, tag, dx=None):
            m, layers = model._mlp_desc(seq, ctx.drop_seed, 100 if tag == "W_p" else 200)
            m.bn_use_running = ctx.bn_running
            m.math_mode = ctx.math_field
            g = L.pn_mlp_grads()
            for i, (lin, bn) in enumerate(layers):
                g.dw[i] = gbuf(lin.weight)
                if bn is not None:
                    g.dgamma[i] = gbuf(bn.weight)
                    g.dbeta[i] = gbuf(bn.bias)
            rows = x.shape[0]
            sv = _save_buf(model, tag, 0, dev, private=ctx.own_save)
            w = L.workspace(lib.pn_mlp_rows_train_ws_bytes(C.byref(m), rows), dev, "train")
            L.check(lib.pn_mlp_rows_bwd(C.byref(m), L.ptr(x), x.shape[1], rows, L.ptr(dy), C.byref(g), L.ptr(dx),
                                        L.ptr(sv), sv.numel(), L.ptr(w), w.numel(), st))

        dP_f = torch.empty_like(P_f) if ctx.needs_input_grad[1] else None  # TRAIN_SEQUENCE_ENCODER: True
        dL_f = torch.empty_like(L_f) if ctx.needs_input_grad[2] else None  # attention-pooled labels: scorer is trained
        mlp_bwd(model.W_p, P_f, dP_e, "W_p", dP_f)
        mlp_bwd(model.W_l, L_f, dL_e, "W_l", dL_f)

        outs = []
        for p, need in zip(ctx.param_list, ctx.needs_input_grad[4:]):
            outs.append(grads.get(id(p)) if need else None)
        ctx.model = None
        ctx.own_save = None  # a private store goes back to the allocator with its backward
        return (None, dP_f, dL_f, None, *outs)


def head_parameters(model):
    from .ProtNote import _split_layers

    ps = _stack_params(_split_layers(model.W_p)) + _stack_params(_split_layers(model.W_l))
    if model.feature_fusion.startswith("concatenation"):
        ps += _stack_params(_split_layers(model.output_layer))
    return ps


def trainable_parameters(model):
    """Every parameter the reference's optimiser would hold for this model besides the encoders
    (ProtNoteTrainer.py:204-231): the heads, plus raw_attn_scorer with LABEL_EMBEDDING_POOLING_METHOD: all."""
    ps = head_parameters(model)
    if getattr(model, "label_embedding_pooling_method", "mean") == "all":
        ps = ps + [model.raw_attn_scorer.weight, model.raw_attn_scorer.bias]
    return ps


class _AttnPoolFn(torch.autograd.Function):
    """Differentiable ProtNote.additive_attention (reference ProtNote.py:154-166) wrt the scorer's weight and bias; the
    token embeddings come from the frozen label encoder (cached), so no gradient flows into them."""

    @staticmethod
    def forward(ctx, model, hidden, mask, weight, bias):
        out = model.additive_attention(hidden, mask)
        ctx.model = model
        ctx.save_for_backward(hidden, mask.to(device=hidden.device, dtype=torch.int64).contiguous(), weight, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        hidden, mask, weight, bias = ctx.saved_tensors
        lib = L.lib()
        N, T, d = hidden.shape
        dout = dout.contiguous().float()
        dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
        db = torch.empty_like(bias)
        ws = L.workspace(lib.pn_additive_attention_bwd_ws_bytes(N, d), hidden.device, "attn")
        L.check(lib.pn_additive_attention_bwd(L.ptr(hidden), L.ptr(mask), L.ptr(weight.detach()), L.ptr(bias.detach()),
                                              L.ptr(dout), N, T, d, L.ptr(dw), L.ptr(db), L.ptr(ws), ws.numel(),
                                              L.stream_ptr()))
        return None, None, None, dw, db


class _PassGradFn(torch.autograd.Function):
    """value = `value`, gradient -> `source` unchanged (value = source + something that does not depend on it)."""

    @staticmethod
    def forward(ctx, source, value):
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _EnsembleFn(torch.autograd.Function):
    """Inference-time description ensembling (ProtNote.py:308-322) inside a differentiated eval-mode forward."""

    @staticmethod
    def forward(ctx, logits, ndesc):
        from .ProtNote import ProtNote

        logits = logits.contiguous()
        ctx.save_for_backward(logits)
        ctx.ndesc = ndesc
        return ProtNote._ensemble(logits, logits.shape[0], logits.shape[1], ndesc, protein_major=True)

    @staticmethod
    def backward(ctx, dout):
        (logits,) = ctx.saved_tensors
        B, NL = logits.shape
        dx = torch.empty_like(logits)
        L.check(L.lib().pn_ensemble_logit_bwd(L.ptr(logits), L.ptr(dout.contiguous().float()), B, NL, ctx.ndesc, L.ptr(dx),
                                              L.stream_ptr()))
        return dx, None


def ensemble_logits(logits, ndesc):
    return _EnsembleFn.apply(logits, int(ndesc))


def forward_train(model, sequence_onehots, sequence_embeddings, sequence_lengths, L_f, label_token_counts,
                  attention_mask=None, opts=None):
    """Reference ProtNote.forward (ProtNote.py:219-309) on the activation-storing kernels: training mode (with or without
    autograd: under torch.no_grad() BatchNorm still takes batch statistics and advances its buffers) and eval mode with
    autograd on (BatchNorm on its running statistics, no noise, no dropout; the result is differentiable)."""
    L_src = L_f
    with torch.no_grad():
        L_f = L_f.detach().float().contiguous()
        if model.training and label_token_counts is not None and model.label_embedding_noising_alpha > 0:
            # (for [N, T, d] token embeddings the reference's scale is alpha / sqrt(L_f.shape[1]) = alpha / sqrt(T),
            #  ProtNote.py:227-230 - _noised reads shape[1] the same way)
            L_f = model._noised(L_f)
    if L_src.requires_grad and torch.is_grad_enabled():
        # the reference uses the caller's tensor as is (ProtNote.py:192-196; the noise is additive): gradients reach it
        L_f = _PassGradFn.apply(L_src, L_f)
    if attention_mask is not None:  # LABEL_EMBEDDING_POOLING_METHOD: all - pooling after the noise (:266-267)
        sc = model.raw_attn_scorer
        L_f = _AttnPoolFn.apply(model, L_f, attention_mask, sc.weight, sc.bias)
    P_f = None
    if sequence_embeddings is not None and (not model.train_sequence_encoder or not model.training):
        P_f = sequence_embeddings.float().contiguous()  # not detached, as in the reference (:243-247)
    elif sequence_onehots is not None and sequence_lengths is not None:
        if model.train_sequence_encoder and model.training:
            # reference ProtNote.py:248-256: encoder inside the autograd graph (differentiable when its
            # parameters require grad: _EncoderTrainFn)
            P_f = model.sequence_encoder.get_embeddings(sequence_onehots, sequence_lengths)
        else:
            with torch.no_grad():
                P_f = model.sequence_encoder.get_embeddings(sequence_onehots, sequence_lengths)
    else:
        raise ValueError("Incompatible sequence parameters passed to forward method.")
    L.require_hip(P_f, L_f)
    L_f = L_f.contiguous()
    if model.training:
        # torch's BatchNorm1d refuses to take batch statistics over ONE row (torch/nn/functional.py _verify_batch_size), so the
        # reference raises ValueError for a single protein / a single label row in training mode (W_p / W_l carry BatchNorm
        # whenever PROJECTION_HEAD_NUM_LAYERS > 1; the output MLP's BatchNorm sees B * N_L rows).  Same exception here instead
        # of a silent variance of zero.
        from .ProtNote import _split_layers

        def _has_bn(seq):
            return any(bn is not None for _, bn in _split_layers(seq))

        for rows, seq in ((P_f.shape[0], model.W_p), (L_f.shape[0], model.W_l)):
            if rows == 1 and _has_bn(seq):
                w = next(lin.out_features for lin, bn in _split_layers(seq) if bn is not None)
                raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([1, {w}])")
        if (model.feature_fusion.startswith("concatenation") and P_f.shape[0] * L_f.shape[0] == 1
                and _has_bn(model.output_layer)):
            w = next(lin.out_features for lin, bn in _split_layers(model.output_layer) if bn is not None)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([1, {w}])")
    # SEQUENCE_EMBEDDING_DROPOUT / LABEL_EMBEDDING_DROPOUT (ProtNote.py:83-86): Bernoulli masks on the [B, 1100] and
    # [N_L, 1024] input rows (after the label noise, as the wrapped W_l sees them); torch's device RNG, like the noise
    from .ProtNote import input_dropout_p

    p_seq, p_lab = input_dropout_p(model.W_p), input_dropout_p(model.W_l)
    if p_seq > 0 and model.training:
        P_f = torch.nn.functional.dropout(P_f, p_seq, training=True)
    if p_lab > 0 and model.training:
        L_f = torch.nn.functional.dropout(L_f, p_lab, training=True)
    opts = {} if opts is None else opts  # "want_embeddings" in, "embeddings" out (see _HeadsTrainFn.forward)
    opts["want_graph"] = torch.is_grad_enabled()
    return _HeadsTrainFn.apply(model, P_f, L_f, opts, *head_parameters(model))
