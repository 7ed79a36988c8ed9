"""ProtNote two-tower fusion model on MI355X - drop-in twin of protnote/models/ProtNote.py (reference :9-378).

Same constructor kwargs (including the reference's misspelt `outout_mlp_add_batchnorm`), same
`forward(...) -> (logits, embeddings_dict)` contract, same state_dict keys (W_p.{0,1,4,5,...},
W_l.*, output_layer.*, sequence_encoder.*).  torch.nn modules are parameter containers only; the
arithmetic runs in libprotnote_hip.so:

  * W_p / W_l row MLPs                -> pn_mlp_rows_fwd_eval / train path (f32-MFMA GEMMs, BN folded
                                         into the next layer's operand load)
  * _get_joint_embeddings + output_layer (reference :112-152, :286-293) -> pn_pairhead_*: the
    [B*N_L, 2d] joint tensor is never built; layer 1 is separable (z1[i,j] = A[i] + Bm[j]) and the
    remaining layers are tiled GEMMs over the pair grid with ReLU/BN fused into operand generation.
  * similarity head (reference :281-284) -> pn_similarity_fwd
  * inference-time description ensembling (reference :308-322) -> pn_ensemble_logit
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib as L

_FUSION_ID = {"concatenation": 0, "concatenation_diff": 1, "concatenation_prod": 2}


def _row_mlp(in_channels, hidden_channels, bias, dropout):
    """Layer order of torchvision.ops.MLP(norm_layer=BatchNorm1d) as built at reference ProtNote.py:63-81:
    (Linear, BatchNorm1d, ReLU, Dropout) per hidden width, then Linear, Dropout - this fixes the
    checkpoint keys W_*.{0,1,4,5,8,9,12}."""
    layers = []
    d = in_channels
    for h in hidden_channels[:-1]:
        layers += [nn.Linear(d, h, bias=bias), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(dropout)]
        d = h
    layers += [nn.Linear(d, hidden_channels[-1], bias=bias), nn.Dropout(dropout)]
    return nn.Sequential(*layers)


def get_mlp(input_dim, hidden_dim, num_layers, input_dropout=0.0, dropout=0.0, batch_norm=False,
            output_neuron_bias=None):
    """Container with the layer order of reference get_mlp (ProtNote.py:337-378)."""
    layers = []
    if input_dropout > 0:
        layers.append(nn.Dropout(input_dropout))
    for idx in range(num_layers):
        layers.append(nn.Linear(input_dim if idx == 0 else hidden_dim, hidden_dim, bias=not batch_norm))
        if batch_norm:
            layers.append(nn.BatchNorm1d(hidden_dim))
        layers.append(nn.ReLU())
        if idx < num_layers - 1:
            layers.append(nn.Dropout(dropout))
    out = nn.Linear(hidden_dim, 1)
    if output_neuron_bias is not None:
        out.bias.data.fill_(output_neuron_bias)
    layers.append(out)
    return nn.Sequential(*layers)


def _bn_struct(bn):
    if bn is None:
        return L.pn_bn(None, None, None, None)
    return L.pn_bn(bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr())


def _split_layers(seq):
    """[(Linear, BatchNorm1d|None), ...] from a Sequential container (Dropout/ReLU skipped)."""
    flat = []

    def walk(m):
        if isinstance(m, nn.Sequential):
            for c in m:
                walk(c)
        else:
            flat.append(m)

    walk(seq)
    out = []
    for m in flat:
        if isinstance(m, nn.Linear):
            out.append([m, None])
        elif isinstance(m, nn.BatchNorm1d):
            out[-1][1] = m
        # nn.Dropout entries are containers only: a Dropout in FRONT of the first Linear is the embedding dropout of
        # reference ProtNote.py:83-86 (applied to the input rows by the train path, see input_dropout_p); the ones
        # BETWEEN the layers (OUTPUT_MLP_DROPOUT) are generated inside the kernels from ProtNote.mlp_dropout and a
        # per-forward seed (pn_mlp.dropout_* / pn_pairhead.dropout_*)
    return out


def input_dropout_p(seq) -> float:
    """p of the nn.Dropout that SEQUENCE_EMBEDDING_DROPOUT / LABEL_EMBEDDING_DROPOUT put in front of W_p / W_l."""
    if isinstance(seq, nn.Sequential) and len(seq) > 0 and isinstance(seq[0], nn.Dropout):
        return float(seq[0].p)
    return 0.0


class ProtNote(nn.Module):
    _warned_eval_stored = False  # the eval-mode + autograd path warns once per process (see forward)
    # Arithmetic of THIS model's big GEMMs: None = the process default at call time (protnote_amd.set_math_mode /
    # set_backward_math / set_forward_math), or "f32" | "bf16x3" and "same" | "bf16".  Carried in every descriptor this model
    # builds (pn_*.math_mode / pn_pairhead.backward_math / .forward_math), so models driven from different host threads can
    # differ.  forward_math = "bf16" + backward_math = "bf16" is the reference's autocast arithmetic class
    # (ProtNoteTrainer.py:287,728-738) for the pair-grid GEMMs of the output MLP; everything else follows math_mode.
    _math_mode = None
    backward_math = None
    forward_math = None

    @property
    def math_mode(self):
        return self._math_mode

    @math_mode.setter
    def math_mode(self, mode):
        if mode is not None:
            L.math_field(mode)  # raises ValueError on anything but "f32" / "bf16x3"
        self.__dict__["_math_mode"] = mode
        enc = self.__dict__.get("_modules", {}).get("sequence_encoder")
        if enc is not None:
            enc.math_mode = mode

    def __init__(self, protein_embedding_dim=1100, label_embedding_dim=1024, label_embedding_pooling_method="mean",
                 inference_descriptions_per_label=1, latent_dim=1024, label_encoder=None, sequence_encoder=None,
                 label_encoder_num_trainable_layers=False, train_sequence_encoder=False,
                 output_mlp_hidden_dim_scale_factor=1024, output_mlp_num_layers=2, output_neuron_bias=None,
                 outout_mlp_add_batchnorm=True, residual_connection=False, dropout=0.0,
                 sequence_embedding_dropout=0.0, label_embedding_dropout=0.0, label_embedding_noising_alpha=0.0,
                 projection_head_num_layers=1, projection_head_hidden_dim_scale_factor=1,
                 label_batch_size_limit=float("inf"), sequence_batch_size_limit=float("inf"),
                 feature_fusion="concatenation", temperature=0.07):
        super().__init__()
        self.label_encoder_num_trainable_layers = label_encoder_num_trainable_layers
        self.train_sequence_encoder = train_sequence_encoder
        self.label_encoder, self.sequence_encoder = label_encoder, sequence_encoder
        self.inference_descriptions_per_label = inference_descriptions_per_label
        self.label_batch_size_limit, self.sequence_batch_size_limit = label_batch_size_limit, sequence_batch_size_limit
        self.feature_fusion = feature_fusion
        self.temperature = temperature
        self.label_embedding_pooling_method = label_embedding_pooling_method
        self.latent_dim = latent_dim
        self.label_embedding_noising_alpha = label_embedding_noising_alpha
        self.residual_connection = residual_connection
        # OUTPUT_MLP_DROPOUT (bin/main.py:421 -> `dropout`): p of the Dropout layers inside W_p, W_l and output_layer
        self.mlp_dropout = float(dropout)
        if not 0.0 <= self.mlp_dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")

        hidden = [latent_dim * projection_head_hidden_dim_scale_factor] * (projection_head_num_layers - 1) + [latent_dim]
        self.W_p = _row_mlp(protein_embedding_dim, hidden, bias=False, dropout=dropout)
        self.W_l = _row_mlp(label_embedding_dim, hidden, bias=False, dropout=dropout)
        # reference :83-86 - wrapping renames the checkpoint keys, kept for compatibility
        if sequence_embedding_dropout > 0:
            self.W_p = nn.Sequential(nn.Dropout(sequence_embedding_dropout), self.W_p)
        if label_embedding_dropout > 0:
            self.W_l = nn.Sequential(nn.Dropout(label_embedding_dropout), self.W_l)
        if self.label_embedding_pooling_method == "all":
            self.raw_attn_scorer = nn.Linear(label_embedding_dim, 1, bias=True)
        if self.feature_fusion.startswith("concatenation"):
            self.output_layer = get_mlp(
                input_dim=self._get_concatenated_features_dim(),
                hidden_dim=int(round(output_mlp_hidden_dim_scale_factor * latent_dim)),
                num_layers=output_mlp_num_layers, output_neuron_bias=output_neuron_bias,
                batch_norm=outout_mlp_add_batchnorm, dropout=dropout)
        # label-chunk size of the eval pair head (rows = chunk * B); None = auto (~512k pair rows)
        self.pair_label_chunk = None
        # eval-mode cache of L_e = W_l(label table) (see _label_projection_eval) and named tables (set_label_table)
        self.__dict__["_pn_le_cache"] = {}
        self.__dict__["_pn_label_tables"] = {}
        self.label_projection_cache_size = 4

    def _get_concatenated_features_dim(self):
        dim = {"concatenation_diff": self.latent_dim * 3, "concatenation_prod": self.latent_dim * 3,
               "concatenation": self.latent_dim * 2}
        return dim[self.feature_fusion]

    # ------------------------------------------------------------------ descriptors
    def _mlp_desc(self, seq, drop_seed=None, drop_stream=0):
        """`drop_seed` (training forward / its backward): enables the in-kernel Dropout(self.mlp_dropout) of this stack
        with that seed; eval descriptors leave it off."""
        layers = _split_layers(seq)
        if len(layers) > L.PN_MAX_LAYERS:
            raise ValueError(f"PROJECTION_HEAD_NUM_LAYERS: at most {L.PN_MAX_LAYERS} projection layers are supported, got {len(layers)}")
        m = L.pn_mlp()
        m.nlayers = len(layers)
        m.dims[0] = layers[0][0].in_features
        eps, mom = 1e-5, 0.1
        for i, (lin, bn) in enumerate(layers):
            m.dims[i + 1] = lin.out_features
            m.w[i] = lin.weight.data_ptr()
            m.bias[i] = lin.bias.data_ptr() if lin.bias is not None else None
            m.bn[i] = _bn_struct(bn)
            if bn is not None:
                eps, mom = bn.eps, bn.momentum
        m.bn_eps, m.bn_momentum = eps, mom
        m.math_mode = L.math_field(self.math_mode)
        if drop_seed is not None and self.mlp_dropout > 0:
            m.dropout_p, m.dropout_seed, m.dropout_stream = self.mlp_dropout, int(drop_seed), int(drop_stream)
        return m, layers

    def _pair_desc(self, drop_seed=None):
        layers = _split_layers(self.output_layer)
        hidden, out = layers[:-1], layers[-1][0]
        if len(hidden) > L.PN_MAX_LAYERS:
            raise ValueError(f"OUTPUT_MLP_NUM_LAYERS: at most {L.PN_MAX_LAYERS} hidden layers are supported, got {len(hidden)}")
        hd = L.pn_pairhead()
        hd.d = self.latent_dim
        hd.in_dim = hidden[0][0].in_features
        if self.feature_fusion not in _FUSION_ID:
            raise NotImplementedError(f"feature_fusion={self.feature_fusion!r} is not implemented in protnote_amd yet")
        hd.fusion = _FUSION_ID[self.feature_fusion]
        hd.nlayers = len(hidden)
        hd.h = hidden[0][0].out_features
        eps, mom = 1e-5, 0.1
        for i, (lin, bn) in enumerate(hidden):
            hd.w[i] = lin.weight.data_ptr()
            hd.bias[i] = lin.bias.data_ptr() if lin.bias is not None else None
            hd.bn[i] = _bn_struct(bn)
            if bn is not None:
                eps, mom = bn.eps, bn.momentum
        hd.w_out = out.weight.data_ptr()
        hd.b_out = out.bias.data_ptr()
        hd.bn_eps, hd.bn_momentum = eps, mom
        hd.math_mode = L.math_field(self.math_mode)
        hd.backward_math = L.backward_math_field(self.backward_math)
        hd.forward_math = L.forward_math_field(self.forward_math)
        if drop_seed is not None and self.mlp_dropout > 0:
            hd.dropout_p, hd.dropout_seed = self.mlp_dropout, int(drop_seed)
        return hd, layers

    # ------------------------------------------------------------------ HIP stages (eval)
    def _project_eval(self, seq, x):
        m, keep = self._mlp_desc(seq)
        rows = x.shape[0]
        lib = L.lib()
        ws = L.workspace(lib.pn_mlp_rows_ws_bytes(C.byref(m), rows), x.device, "mlp")
        y = torch.empty(rows, m.dims[m.nlayers], dtype=torch.float32, device=x.device)
        L.check(lib.pn_mlp_rows_fwd_eval(C.byref(m), L.ptr(x), x.shape[1], rows, L.ptr(y), L.ptr(ws), ws.numel(),
                                         L.stream_ptr()))
        del keep
        return y

    # ------------------------------------------------------------------ eval-mode label projection cache
    def train(self, mode=True):
        """A train-mode forward moves W_l's BatchNorm buffers and an optimiser step its weights through raw device
        pointers (no tensor version changes), so every train()/eval() switch drops the cached label projections."""
        self.__dict__.setdefault("_pn_le_cache", {}).clear()
        return super().train(mode)

    def _w_l_state_key(self):
        from ..utils.optim import weights_generation

        ts = list(self.W_l.parameters()) + list(self.W_l.buffers())
        if any(t.is_inference() for t in ts):  # no version counters (model built under torch.inference_mode()): no cache
            return None
        vs = tuple((t.data_ptr(), t._version) for t in ts)
        return (weights_generation(), L.math_field(self.math_mode), vs)

    def _label_projection_eval(self, label_embeddings):
        """L_e = W_l(L_f) in eval mode depends only on the label table and W_l (reference ProtNote.py:192-196,270-271
        recomputes it on every call: 3.2 TFLOP for 64 204 description rows, i.e. 1/B of the head's work - 12 % at the
        reference's per-GPU batch of 8).  Cached per table OBJECT: the entry holds a reference to the tensor it was
        computed from (so its address cannot be recycled) and is valid while that tensor's version counter, W_l's
        parameters / buffers (addresses + version counters), the fused optimiser's step generation and the math mode are
        unchanged; train()/eval() switches clear it.  A fresh tensor per call simply recomputes, as the reference does."""
        cache = self.__dict__.setdefault("_pn_le_cache", {})
        t = label_embeddings
        if int(self.label_projection_cache_size) <= 0 or t.is_inference():
            # cache off: recompute per call, like the reference.  Tensors created under torch.inference_mode() carry no
            # version counter, so an in-place edit of such a table could not be noticed: never cached
            if int(self.label_projection_cache_size) <= 0:
                cache.clear()
            return self._project_eval(self.W_l, t.detach().float().contiguous())
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)
        wkey = self._w_l_state_key()
        if wkey is None:
            return self._project_eval(self.W_l, t.detach().float().contiguous())
        state = (t._version, wkey)
        hit = cache.get(key)
        if hit is not None and hit[0] is t and hit[1] == state:
            cache[key] = cache.pop(key)  # most recently used last
            return hit[2]
        L_e = self._project_eval(self.W_l, t.detach().float().contiguous())
        cache.pop(key, None)
        cache[key] = (t, state, L_e)
        while len(cache) > max(int(self.label_projection_cache_size), 0):
            cache.pop(next(iter(cache)))
        return L_e

    def set_label_table(self, name, label_embeddings):
        """Register a label table (cached description embeddings [N_L * n_desc, d_l] on the device) under `name`;
        forward(label_embeddings=name) then scores against it.  Swapping GO -> EC at run time (BASELINE configs[4],
        reference bin/test_models.py:14-23 re-runs main.py per test set) is a dictionary lookup: the table stays in HBM
        and its projection L_e is computed once per weight state, not once per batch."""
        L.require_hip(label_embeddings)
        self.__dict__.setdefault("_pn_label_tables", {})[str(name)] = label_embeddings
        return label_embeddings

    def label_table(self, name):
        try:
            return self.__dict__.get("_pn_label_tables", {})[str(name)]
        except KeyError:
            raise KeyError(f"no label table named {name!r}; call set_label_table first") from None

    def _needs_graph(self, sequence_embeddings, label_embeddings) -> bool:
        """Would autograd record anything for this forward?  (A trainable head / scorer parameter, or an input that
        requires grad; the encoder only contributes in training mode, ProtNote.py:248-264.)"""
        from .train_path import trainable_parameters

        if any(p.requires_grad for p in trainable_parameters(self)):
            return True
        return any(t is not None and torch.is_tensor(t) and t.requires_grad for t in (sequence_embeddings, label_embeddings))

    def _train_chunk(self, B, NL):
        """Labels per chunk of the backward ring (rows = chunk * B); ~256k pair rows by default (each chunk is one
        GEMM launch of ~49k workgroups, so launch tails are <1 %; costs one chunk of slack per stored layer)."""
        if self.pair_label_chunk:
            return int(self.pair_label_chunk)
        return max(1, min(NL, (256 * 1024) // max(B, 1)))

    def _auto_chunk(self, B, NL):
        if self.pair_label_chunk:
            return int(self.pair_label_chunk)
        return max(1, min(NL, (512 * 1024) // max(B, 1)))

    def _pairhead_eval(self, P_e, L_e):
        hd, keep = self._pair_desc()
        B, NL = P_e.shape[0], L_e.shape[0]
        chunk = self._auto_chunk(B, NL)
        lib = L.lib()
        ws = L.workspace(lib.pn_pairhead_eval_ws_bytes(C.byref(hd), B, NL, chunk), P_e.device, "pair")
        pairs = torch.empty(NL * B, dtype=torch.float32, device=P_e.device)
        L.check(lib.pn_pairhead_fwd_eval(C.byref(hd), L.ptr(P_e), L.ptr(L_e), B, NL, L.ptr(pairs), chunk,
                                         L.ptr(ws), ws.numel(), L.stream_ptr()))
        del keep
        return pairs  # label-major pair grid: pairs[j*B + i]

    def _pairhead_eval_with_embeddings(self, P_e, L_e):
        """save_embeddings=True (reference ProtNote.py:294-302,326-332): besides the logits, return the joint
        embeddings and the penultimate output-MLP activations as CPU tensors in the reference's protein-major
        row order (row = i * N_L + j).  Small evaluation subsets only: [B*N_L, 3072] is materialised."""
        hd, keep = self._pair_desc()
        B, NL = P_e.shape[0], L_e.shape[0]
        lib = L.lib()
        ws = L.workspace(lib.pn_pairhead_hidden_ws_bytes(C.byref(hd), B, NL), P_e.device, "pair")
        pairs = torch.empty(NL * B, dtype=torch.float32, device=P_e.device)
        hidden = torch.empty(NL, B, hd.h, dtype=torch.float32, device=P_e.device)
        L.check(lib.pn_pairhead_fwd_eval_hidden(C.byref(hd), L.ptr(P_e), L.ptr(L_e), B, NL, L.ptr(pairs), L.ptr(hidden),
                                                L.ptr(ws), ws.numel(), L.stream_ptr()))
        del keep
        out_emb = hidden.permute(1, 0, 2).reshape(B * NL, hd.h).cpu()
        return pairs, {"output_layer_embeddings": out_emb, "joint_embeddings": self._joint_embeddings_cpu(P_e, L_e)}

    def _get_joint_embeddings(self, P_e, L_e, num_sequences, num_labels):
        """Reference ProtNote._get_joint_embeddings (ProtNote.py:112-152), same signature: the [num_sequences * num_labels,
        2d | 3d] joint tensor, protein-major rows i * num_labels + j, on the inputs' device.  The kernels never build it
        (layer 1 is separable) - this is the layout contract for callers that ask for the tensor itself
        (`save_embeddings`, ProtNote.py:324-332); data movement only, no arithmetic beyond the reference's own diff / product."""
        d = P_e.shape[1]
        joint = torch.cat([P_e.unsqueeze(1).expand(-1, num_labels, -1), L_e.unsqueeze(0).expand(num_sequences, -1, -1)], dim=2)
        joint = joint.reshape(-1, joint.shape[-1])
        if self.feature_fusion == "concatenation_diff":
            joint = torch.cat([joint, joint[:, :d] - joint[:, d:]], dim=-1)
        elif self.feature_fusion == "concatenation_prod":
            joint = torch.cat([joint, joint[:, :d] * joint[:, d:]], dim=-1)
        return joint

    def _joint_embeddings_cpu(self, P_e, L_e):
        """The reference's joint tensor as save_embeddings returns it (ProtNote.py:326-328: detached, on the CPU)."""
        pe, le = P_e.detach().cpu(), L_e.detach().cpu()
        return self._get_joint_embeddings(pe, le, pe.shape[0], le.shape[0])

    def additive_attention(self, hidden_states, attention_mask):
        """Reference ProtNote.additive_attention (ProtNote.py:154-166), inference: masked-softmax attention pooling of
        token embeddings [N, T, d] with the raw_attn_scorer Linear(d, 1)."""
        L.require_hip(hidden_states, attention_mask)
        hs = hidden_states.detach().float().contiguous()
        mask = attention_mask.to(device=hs.device, dtype=torch.int64).contiguous()
        N, T, d = hs.shape
        out = torch.empty(N, d, dtype=torch.float32, device=hs.device)
        L.check(L.lib().pn_additive_attention(L.ptr(hs), L.ptr(mask), L.ptr(self.raw_attn_scorer.weight.detach()),
                                              L.ptr(self.raw_attn_scorer.bias.detach()), N, T, d, L.ptr(out),
                                              L.stream_ptr()))
        return out

    def _similarity(self, P_e, L_e):
        B, NL, d = P_e.shape[0], L_e.shape[0], P_e.shape[1]
        lib = L.lib()
        ws = L.workspace(lib.pn_similarity_ws_bytes(B, NL), P_e.device, "sim")
        logits = torch.empty(B, NL, dtype=torch.float32, device=P_e.device)
        L.check(lib.pn_similarity_fwd(L.ptr(P_e), L.ptr(L_e), B, NL, d, float(self.temperature), L.ptr(logits),
                                      L.ptr(ws), ws.numel(), L.stream_ptr()))
        return logits

    @staticmethod
    def _ensemble(pairs, B, NL, ndesc, protein_major=False):
        out = torch.empty(B, NL // ndesc, dtype=torch.float32, device=pairs.device)
        L.check(L.lib().pn_ensemble_logit(L.ptr(pairs), B, NL, ndesc, 1 if protein_major else 0, L.ptr(out),
                                          L.stream_ptr()))
        return out

    # Where the uniforms of the label-embedding noise (ProtNote.py:219-240) come from: "kernel" (default) = a counter hash of
    # (seed, row, column) evaluated inside pn_label_noise_seeded - the seed is one draw from torch's host generator per forward,
    # so torch.manual_seed governs it, and no [N_L, d] tensor of uniforms is written and read back; "torch" = torch.rand_like
    # on the device, the reference's own call (what tests that replay a reference run's noise hook into).
    label_noise_rng = "kernel"

    def _noised(self, L_f, u=None):
        """Reference :219-240: L_f + (2u - 1) * alpha / sqrt(L_f.shape[1]), u ~ U[0,1).  `u` given: that draw; else by
        `label_noise_rng`.  The seed of the last in-kernel draw is kept in `_pn_last_noise_seed` (pn_uniform reproduces it)."""
        out = torch.empty_like(L_f)
        scale = float(self.label_embedding_noising_alpha) / math.sqrt(L_f.shape[1])
        if u is None and self.label_noise_rng == "torch":
            u = torch.rand_like(L_f)
        if u is not None:
            L.check(L.lib().pn_label_noise(L.ptr(L_f), L.ptr(u.contiguous()), scale, L.ptr(out), L_f.numel(),
                                           L.stream_ptr()))
            return out
        if self.label_noise_rng != "kernel":
            raise ValueError(f"label_noise_rng must be 'kernel' or 'torch', got {self.label_noise_rng!r}")
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())  # host generator: no device sync
        self.__dict__["_pn_last_noise_seed"] = seed
        cols = int(L_f.shape[-1])
        L.check(L.lib().pn_label_noise_seeded(L.ptr(L_f), seed, scale, L.ptr(out), L_f.numel() // cols, cols, L.stream_ptr()))
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, sequence_onehots=None, sequence_embeddings=None, sequence_lengths=None, tokenized_labels=None,
                label_embeddings=None, label_token_counts=None, save_embeddings=False):
        """Reference ProtNote.forward (ProtNote.py:168-334).  Returns (logits [B, N_L], embeddings dict)."""
        # ---- label branch (:192-217): cached-embedding path only ----
        if isinstance(label_embeddings, str):  # a table registered with set_label_table
            label_embeddings = self.label_table(label_embeddings)
        if label_embeddings is not None and (self.label_encoder_num_trainable_layers == 0 or not self.training):
            L_f = label_embeddings
        elif tokenized_labels is not None and self.training:
            raise NotImplementedError("on-the-fly label encoding (tokenized_labels) is outside the MI355X hot path; "
                                      "pass cached label_embeddings")
        else:
            raise ValueError("Incompatible label parameters passed to forward method.")
        L.require_hip(L_f)
        pool_all = self.label_embedding_pooling_method == "all"
        attn_mask = None
        # save_embeddings (ProtNote.py:292-302,324-332; the trainer passes the flag through, ProtNoteTrainer.py:288) is accepted
        # in every mode, as in the reference: with `similarity` there is nothing to save (both lists stay empty); with the
        # concatenation heads the joint tensor and the penultimate output-MLP activations come back detached on the CPU -
        # from the fused inference kernels in eval mode under no_grad, from the activation store otherwise.
        want_embeddings = bool(save_embeddings) and self.feature_fusion.startswith("concatenation")
        # Which kernels run.  The reference puts no restriction on (mode, autograd) combinations (ProtNote.py:168-334):
        #   train mode, autograd on / off -> the activation-storing path (train-mode BatchNorm: batch statistics, buffers
        #                                    advance - also under torch.no_grad());
        #   eval mode, autograd on and something to differentiate -> the same path with BatchNorm on its running
        #                                    statistics (pn_*.bn_use_running): logits are differentiable;
        #   eval mode otherwise           -> the fused inference kernels (nothing stored).
        stored = self.training or (torch.is_grad_enabled() and self._needs_graph(sequence_embeddings, label_embeddings))
        if stored and not self.training and not ProtNote._warned_eval_stored:
            ProtNote._warned_eval_stored = True
            import warnings

            warnings.warn("ProtNote: eval-mode forward with autograd enabled takes the activation-storing path (differentiable "
                          "logits, as in the reference) - it stores B x N_L x h activations per layer and bypasses the fused "
                          "inference kernels and the label-projection cache.  Run inference under torch.no_grad() or "
                          "torch.inference_mode().", stacklevel=2)
        if pool_all:
            # token embeddings [N, T, d] + tokenized_labels["attention_mask"] (ProtNote.py:266-267).  On the stored path
            # the pooling happens inside forward_train (after the label noise, as in the reference, and inside the
            # autograd graph so raw_attn_scorer is trained).
            if tokenized_labels is None or "attention_mask" not in tokenized_labels:
                raise ValueError("LABEL_EMBEDDING_POOLING_METHOD='all' needs tokenized_labels['attention_mask']")
            attn_mask = tokenized_labels["attention_mask"]
            if not stored:
                L_f = self.additive_attention(L_f, attn_mask)

        with torch.autocast(device_type="cuda", enabled=False):  # kernels are f32; ignore AMP (ProtNoteTrainer.py:728)
            if stored:
                from .train_path import ensemble_logits, forward_train

                opts = {"want_embeddings": want_embeddings}  # owned by this call: flag in, embeddings out
                logits = forward_train(self, sequence_onehots, sequence_embeddings, sequence_lengths, L_f,
                                       label_token_counts, attn_mask, opts)
                saved = opts.get("embeddings")
                ndesc = 1 if self.training else int(self.inference_descriptions_per_label)
                if ndesc != 1:  # ProtNote.py:308-322, differentiable
                    logits = ensemble_logits(logits, ndesc)
                if want_embeddings and saved is not None:
                    return logits, saved
                return logits, {"output_layer_embeddings": [], "joint_embeddings": []}

            with torch.no_grad():
                table = L_f if not (pool_all or self.training) else None  # the caller's table object: cacheable
                L_f = L_f.detach().float().contiguous()
                if self.training and label_token_counts is not None and self.label_embedding_noising_alpha > 0:
                    # reference :219-240
                    L_f = self._noised(L_f)
                # ---- sequence branch (:243-264) ----
                if sequence_embeddings is not None and (not self.train_sequence_encoder or not self.training):
                    P_f = sequence_embeddings.detach().float().contiguous()
                elif sequence_onehots is not None and sequence_lengths is not None:
                    P_f = self.sequence_encoder.get_embeddings(sequence_onehots, sequence_lengths)
                else:
                    raise ValueError("Incompatible sequence parameters passed to forward method.")
                P_e = self._project_eval(self.W_p, P_f)
                L_e = self._label_projection_eval(table) if table is not None else self._project_eval(self.W_l, L_f)
                B, NL = P_e.shape[0], L_e.shape[0]
                ndesc = 1 if self.training else int(self.inference_descriptions_per_label)
                if self.feature_fusion == "similarity":
                    logits = self._similarity(P_e, L_e)
                    if ndesc != 1:
                        logits = self._ensemble(logits, B, NL, ndesc, protein_major=True)
                elif self.feature_fusion.startswith("concatenation"):
                    if want_embeddings:
                        pairs, embeddings = self._pairhead_eval_with_embeddings(P_e, L_e)
                        return self._ensemble(pairs, B, NL, ndesc), embeddings
                    pairs = self._pairhead_eval(P_e, L_e)
                    logits = self._ensemble(pairs, B, NL, ndesc)
                else:
                    raise ValueError("feature fusion method not implemented")
        return logits, {"output_layer_embeddings": [], "joint_embeddings": []}
