"""CPU oracle for the ProtNote forward/training hot path.   *** TEST INFRASTRUCTURE, NOT PRODUCT ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(protnote_amd/) never calls it and fails loudly when the HIP library is missing.

A functional, float32, torch-CPU restatement of the reference algorithm operating directly on a
reference-style ``state_dict`` (dict name -> tensor).  It deliberately keeps the reference's *naive*
formulation (materialised [B*N_L, 2d] joint tensor, explicit padding masks) so that it is an
independent check of the factorised HIP implementation.  All line numbers cite /root/reference.

Parity pin: checked against golden vectors produced by running the reference itself on CPU
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).  The reference has
no golden vectors / KATs of its own for this path.  torchvision.ops.MLP (third-party,
torchvision==0.15.2 per the reference's setup.py:17, absent here) is restated as its published layer
order Linear,[norm],ReLU,Dropout ... Linear,Dropout; that ordering is pinned by the state-dict key
names W_p.{0,1,4,5,8,9,12} only.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ------------------------------------------------------------------------------------------------
# encoder  (protnote/models/protein_encoders.py, protnote/data/datasets.py:535-569)
# ------------------------------------------------------------------------------------------------
def set_padding_to_sentinel(x: Tensor, lens: Tensor, sentinel: float) -> Tensor:
    """datasets.py:535-569 - x[b, :, t] = sentinel for t >= lens[b]; x is [B, C, L]."""
    b, _, l = x.shape
    pad = torch.arange(l, device=x.device).expand(b, l) >= lens.to(x.device).unsqueeze(1)
    return torch.where(pad.unsqueeze(1).expand_as(x), torch.tensor(sentinel, dtype=x.dtype, device=x.device), x)


def masked_conv1d(x: Tensor, lens: Tensor, w: Tensor, b: Tensor, dilation: int) -> Tensor:
    """protein_encoders.py:8-17 - mask, Conv1d(padding='same', stride 1), mask."""
    x = set_padding_to_sentinel(x, lens, 0.0)
    x = F.conv1d(x, w, b, stride=1, padding="same", dilation=dilation)
    return set_padding_to_sentinel(x, lens, 0.0)


def _bn(x: Tensor, sd: SD, prefix: str, training: bool, eps: float, momentum: float) -> Tensor:
    """torch.nn.BatchNorm1d semantics incl. in-place running-stat update when training."""
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    y = F.batch_norm(x, rm, rv, sd[prefix + "weight"], sd[prefix + "bias"], training, momentum, eps)
    if training and (prefix + "num_batches_tracked") in sd:
        sd[prefix + "num_batches_tracked"] += 1
    return y


def residual_block(x: Tensor, lens: Tensor, sd: SD, p: str, dilation: int, training: bool) -> Tensor:
    """protein_encoders.py:61-67 with BN eps=1e-3, momentum=0.01 (:36,:48)."""
    out = F.relu(_bn(x, sd, p + "bn_activation_1.0.", training, 1e-3, 0.01))
    out = masked_conv1d(out, lens, sd[p + "masked_conv1.weight"], sd[p + "masked_conv1.bias"], dilation)
    out = F.relu(_bn(out, sd, p + "bn_activation_2.0.", training, 1e-3, 0.01))
    out = masked_conv1d(out, lens, sd[p + "masked_conv2.weight"], sd[p + "masked_conv2.bias"], 1)
    return out + x


def num_resnet_blocks(sd: SD, prefix: str = "") -> int:
    n = 0
    while f"{prefix}resnet_blocks.{n}.masked_conv1.weight" in sd:
        n += 1
    return n


def proteinfer_get_embeddings(sd: SD, x: Tensor, lens: Tensor, training: bool = False,
                              dilation_base: int = 3, prefix: str = "",
                              taps: Optional[dict] = None) -> Tensor:
    """protein_encoders.py:109-118.  `sd` buffers are updated in place when training."""
    feats = masked_conv1d(x, lens, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], 1)
    if taps is not None:
        taps["conv1"] = feats
    for i in range(num_resnet_blocks(sd, prefix)):
        feats = residual_block(feats, lens, sd, f"{prefix}resnet_blocks.{i}.", dilation_base ** i, training)
        if taps is not None:
            taps[f"block{i}"] = feats
    feats = set_padding_to_sentinel(feats, lens, 0.0)
    return feats.sum(dim=-1) / lens.unsqueeze(-1)


def proteinfer_forward(sd: SD, x: Tensor, lens: Tensor, training: bool = False, dilation_base: int = 3,
                       prefix: str = "") -> Tensor:
    """protein_encoders.py:120-123."""
    e = proteinfer_get_embeddings(sd, x, lens, training, dilation_base, prefix)
    return F.linear(e, sd[prefix + "output_layer.weight"], sd[prefix + "output_layer.bias"])


# ------------------------------------------------------------------------------------------------
# heads  (protnote/models/ProtNote.py)
# ------------------------------------------------------------------------------------------------
def _linear_indices(sd: SD, prefix: str):
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)
                  and k.endswith(".weight") and sd[k].dim() == 2})
    return idx


def mlp_rows(sd: SD, prefix: str, x: Tensor, training: bool, masks: Optional[dict] = None) -> Tensor:
    """torchvision.ops.MLP as built at ProtNote.py:63-81: (Linear no-bias, BN(eps 1e-5, mom 0.1), ReLU,
    Dropout(p)) x (n-1), Linear no-bias, Dropout(p).  `masks` (training with dropout > 0): the Dropout draws as
    multiplicative tensors keep / (1 - p), keyed f"{prefix}{n}" for hidden layer n and f"{prefix}out" for the output."""
    lin = _linear_indices(sd, prefix)
    for n, i in enumerate(lin):
        x = F.linear(x, sd[f"{prefix}{i}.weight"], sd.get(f"{prefix}{i}.bias"))
        if n < len(lin) - 1:
            x = F.relu(_bn(x, sd, f"{prefix}{i + 1}.", training, 1e-5, 0.1))
            if masks is not None:
                x = x * masks[f"{prefix}{n}"]
        elif masks is not None:
            x = x * masks[f"{prefix}out"]
    return x


def output_mlp(sd: SD, prefix: str, x: Tensor, training: bool, masks: Optional[dict] = None,
               aux: Optional[dict] = None) -> Tensor:
    """get_mlp ProtNote.py:337-378 with batch_norm=True: (Linear no-bias, BN, ReLU[, Dropout]) x n, Linear(h,1); the
    Dropout follows every hidden layer except the last (:369-371).  n >= 1 (OUTPUT_MLP_NUM_LAYERS: 1 = one hidden layer).
    `masks`: keep / (1 - p) tensors keyed f"{prefix}{n}" with rows in the joint tensor's protein-major order.
    `aux` receives "output_layer_embeddings": the input of the output neuron, i.e. what save_embeddings returns
    (ProtNote.py:294-302: the layers are applied one by one up to index len - 2, the last ReLU)."""
    lin = _linear_indices(sd, prefix)
    for n, i in enumerate(lin):
        if n == len(lin) - 1 and aux is not None:
            aux["output_layer_embeddings"] = x
        x = F.linear(x, sd[f"{prefix}{i}.weight"], sd.get(f"{prefix}{i}.bias"))
        if n < len(lin) - 1:
            if f"{prefix}{i + 1}.running_mean" in sd:
                x = _bn(x, sd, f"{prefix}{i + 1}.", training, 1e-5, 0.1)
            x = F.relu(x)
            if masks is not None and n < len(lin) - 2:
                x = x * masks[f"{prefix}{n}"]
    return x


def joint_embeddings(P_e: Tensor, L_e: Tensor, fusion: str) -> Tensor:
    """ProtNote.py:112-152 - row index = i * N_L + j (protein-major)."""
    b, n = P_e.shape[0], L_e.shape[0]
    j = torch.cat([P_e[:, None, :].expand(b, n, -1), L_e[None, :, :].expand(b, n, -1)], dim=2)
    j = j.reshape(b * n, -1)
    d = P_e.shape[1]
    if fusion == "concatenation_diff":
        j = torch.cat([j, j[:, :d] - j[:, d:]], dim=-1)
    if fusion == "concatenation_prod":
        j = torch.cat([j, j[:, :d] * j[:, d:]], dim=-1)
    return j


def noised_label_embeddings(L_f: Tensor, alpha: float, u: Tensor) -> Tensor:
    """ProtNote.py:219-240 with the uniform sample `u` in [0,1) supplied by the caller."""
    return L_f + (2 * u - 1) * (alpha / math.sqrt(L_f.shape[1]))


def additive_attention(sd: SD, hidden_states: Tensor, attention_mask: Tensor) -> Tensor:
    """ProtNote.additive_attention (ProtNote.py:154-166): masked-softmax pooling of token embeddings [N, T, d] with
    the raw_attn_scorer Linear(d, 1)."""
    raw = F.linear(hidden_states, sd["raw_attn_scorer.weight"], sd["raw_attn_scorer.bias"]).squeeze(-1)
    raw = raw.masked_fill(attention_mask == 0, float("-inf"))
    return torch.bmm(torch.softmax(raw, dim=-1).unsqueeze(1), hidden_states).squeeze(1)


def protnote_forward(sd: SD, onehots: Optional[Tensor], lens: Optional[Tensor], label_embeddings: Tensor,
                     *, fusion: str = "concatenation", training: bool = False, temperature: float = 0.07,
                     descriptions_per_label: int = 1, noise_alpha: float = 0.0,
                     noise_u: Optional[Tensor] = None, label_token_counts: Optional[Tensor] = None,
                     dilation_base: int = 3, sequence_embeddings: Optional[Tensor] = None,
                     aux: Optional[dict] = None, train_sequence_encoder: bool = False,
                     attention_mask: Optional[Tensor] = None, dropout_masks: Optional[dict] = None) -> Tensor:
    """ProtNote.forward (ProtNote.py:168-334), cached-label-embedding path; the encoder runs under no_grad unless
    train_sequence_encoder and training (ProtNote.py:248-260).  `attention_mask` given = LABEL_EMBEDDING_POOLING_METHOD
    'all': label_embeddings are token embeddings [N, T, d], pooled AFTER the noise (ProtNote.py:266-267)."""
    L_f = label_embeddings
    if training and label_token_counts is not None and noise_alpha > 0:
        L_f = noised_label_embeddings(L_f, noise_alpha, noise_u)
    if attention_mask is not None:
        L_f = additive_attention(sd, L_f, attention_mask)
    if sequence_embeddings is not None:
        P_f = sequence_embeddings
    elif train_sequence_encoder and training:
        P_f = proteinfer_get_embeddings(sd, onehots, lens, training, dilation_base, "sequence_encoder.")
    else:
        with torch.no_grad():
            P_f = proteinfer_get_embeddings(sd, onehots, lens, training, dilation_base, "sequence_encoder.")
    P_e = mlp_rows(sd, "W_p.", P_f, training, dropout_masks)
    L_e = mlp_rows(sd, "W_l.", L_f, training, dropout_masks)
    if aux is not None:
        aux.update(P_f=P_f, P_e=P_e, L_e=L_e)
    b, n = P_e.shape[0], L_e.shape[0]
    if fusion == "similarity":
        logits = torch.mm(F.normalize(P_e, dim=-1, p=2), F.normalize(L_e, dim=-1, p=2).t()) / temperature
    elif fusion.startswith("concatenation"):
        joint = joint_embeddings(P_e, L_e, fusion)
        if aux is not None:  # save_embeddings (ProtNote.py:324-332)
            aux["joint_embeddings"] = joint
        logits = output_mlp(sd, "output_layer.", joint, training, dropout_masks, aux)
    else:
        raise ValueError("feature fusion method not implemented")
    if training or descriptions_per_label == 1:
        return logits.reshape(b, n)
    p = torch.sigmoid(logits).reshape(b, n // descriptions_per_label, descriptions_per_label).mean(-1)
    return torch.special.logit(p, eps=1e-7)


def train_forward_chunked(sd: SD, P_f: Tensor, L_f: Tensor, *, fusion: str = "concatenation", label_chunk: int = 512,
                          momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """ProtNote.forward in TRAINING mode (ProtNote.py:270-309: W_p, W_l, _get_joint_embeddings :112-152, output_layer
    = get_mlp :337-378 with train-mode BatchNorm1d) for pair grids too large to materialise: the same naive
    formulation - joint rows, Linear, BatchNorm over ALL B*N_L rows, ReLU - evaluated in label chunks.  A layer's
    batch statistics need every row before any row can be normalised, so hidden layer n costs one pass that recomputes
    layers < n from the joint rows; one more pass produces the logits.  Column sums are accumulated in float64;
    running_mean / running_var / num_batches_tracked in `sd` advance exactly like torch's BatchNorm1d (unbiased
    variance into running_var).  Device-agnostic plain torch: tests run it on CPU against protnote_forward (small
    grids) and on the GPU at BASELINE configs[2] size, where it is the independent check of the HIP train forward.
    Returns logits [B, N_L]."""
    P_e = mlp_rows(sd, "W_p.", P_f, True)
    L_e = mlp_rows(sd, "W_l.", L_f, True)
    b, n = P_e.shape[0], L_e.shape[0]
    rows = b * n
    lin = _linear_indices(sd, "output_layer.")
    hidden, out_i = lin[:-1], lin[-1]
    stats = []  # per normalised hidden layer: (mean f32, scale f32, shift f32) or None without BatchNorm

    def chain(j0: int, j1: int, upto: int) -> Tensor:
        """pre-activation of hidden layer `upto` (or the logits for upto == len(hidden)) for labels [j0, j1)."""
        x = joint_embeddings(P_e, L_e[j0:j1], fusion)
        for m, i in enumerate(hidden[:upto]):
            x = F.linear(x, sd[f"output_layer.{i}.weight"], sd.get(f"output_layer.{i}.bias"))
            if stats[m] is not None:
                mean, scale, shift = stats[m]
                x = x.sub_(mean).mul_(scale).add_(shift)
            x = F.relu_(x)
        i = hidden[upto] if upto < len(hidden) else out_i
        return F.linear(x, sd[f"output_layer.{i}.weight"], sd.get(f"output_layer.{i}.bias"))

    for m, i in enumerate(hidden):
        pre = f"output_layer.{i + 1}."
        if pre + "running_mean" not in sd:
            stats.append(None)
            continue
        s1 = torch.zeros(sd[pre + "weight"].shape[0], dtype=torch.float64, device=P_e.device)
        s2 = torch.zeros_like(s1)
        for j0 in range(0, n, label_chunk):
            z = chain(j0, min(n, j0 + label_chunk), m).double()
            s1 += z.sum(0)
            s2 += (z * z).sum(0)
        mean = s1 / rows
        var = (s2 / rows - mean * mean).clamp_min(0.0)  # biased, as used for normalisation
        dt = sd[pre + "weight"].dtype
        scale = (sd[pre + "weight"].double() / torch.sqrt(var + eps)).to(dt)
        stats.append((mean.to(dt), scale, sd[pre + "bias"]))
        sd[pre + "running_mean"].mul_(1 - momentum).add_(mean.to(dt), alpha=momentum)
        sd[pre + "running_var"].mul_(1 - momentum).add_((var * (rows / max(rows - 1, 1))).to(dt), alpha=momentum)
        if pre + "num_batches_tracked" in sd:
            sd[pre + "num_batches_tracked"] += 1
    logits = torch.empty(b, n, dtype=P_e.dtype, device=P_e.device)
    for j0 in range(0, n, label_chunk):
        j1 = min(n, j0 + label_chunk)
        logits[:, j0:j1] = chain(j0, j1, len(hidden)).reshape(b, j1 - j0)
    return logits


def train_grads_chunked(sd: SD, P_f: Tensor, L_f: Tensor, multihots: Tensor, *, fusion: str = "concatenation",
                        label_chunk: int = 512, loss: str = "BCE", momentum: float = 0.1, eps: float = 1e-5,
                        **loss_kw) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """Forward + backward of the train step's heads (ProtNoteTrainer.py:728-738 over ProtNote.py:270-309) for pair grids
    too large to materialise - the backward twin of train_forward_chunked, same naive formulation in label chunks.
    BatchNorm1d's backward over ALL B*N_L rows, dz = gamma/sigma (du - mean(du) - xhat mean(du xhat)), needs the two
    global means of a layer before any row of its dz exists, and du of a layer depends on dz of the layer above: one pass
    per hidden layer, top down, each recomputing the chunk's forward chain and the backward chain above it; a last pass
    accumulates the weight gradients (float64 accumulators) and the gradients wrt P_e / L_e, which then flow through W_p /
    W_l by plain autograd.  Plain torch, device-agnostic: pinned on CPU to train_step's autograd (and through it to the
    reference goldens); on the GPU it is the independent reference of the full-size HIP train step.
    Returns (logits [B, N_L], loss, {parameter name: gradient}) for every trainable head parameter; `sd` buffers advance."""
    head = [k for k in trainable_names(sd) if k.startswith(("W_p.", "W_l."))]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in head}
    work = dict(sd)
    work.update(leaves)
    P_e_g = mlp_rows(work, "W_p.", P_f, True)
    L_e_g = mlp_rows(work, "W_l.", L_f, True)
    P_e, L_e = P_e_g.detach(), L_e_g.detach()
    b, n, d = P_e.shape[0], L_e.shape[0], P_e.shape[1]
    rows = b * n
    dev, dt = P_e.device, P_e.dtype
    lin = _linear_indices(sd, "output_layer.")
    hidden, out_i = lin[:-1], lin[-1]
    nl = len(hidden)
    W = [sd[f"output_layer.{i}.weight"] for i in hidden]
    bias = [sd.get(f"output_layer.{i}.bias") for i in hidden]
    has_bn = [f"output_layer.{i + 1}.running_mean" in sd for i in hidden]
    w_out, b_out = sd[f"output_layer.{out_i}.weight"], sd.get(f"output_layer.{out_i}.bias")
    fstat = [None] * nl  # (mean, invstd, gamma, beta) per BatchNorm

    def forward_chunk(j0, j1, upto, keep=False):
        """-> (x, zs, us): joint rows, pre-BatchNorm z_m and pre-ReLU u_m of hidden layers < upto (kept when asked),
        and the last tensor computed (z of layer `upto`, or the logits)."""
        x = joint_embeddings(P_e, L_e[j0:j1], fusion)
        zs, us = [], []
        h = x
        for m in range(min(upto, nl)):
            z = F.linear(h, W[m], bias[m])
            if has_bn[m]:
                mean, invstd, gamma, beta = fstat[m]
                u = (z - mean) * (invstd * gamma) + beta
            else:
                u = z
            if keep:
                zs.append(z)
                us.append(u)
            h = F.relu(u)
        if upto < nl:
            last = F.linear(h, W[upto], bias[upto])
        else:
            last = F.linear(h, w_out, b_out)
        return x, zs, us, h, last

    # ---- forward statistics, layer by layer (as train_forward_chunked)
    for m in range(nl):
        if not has_bn[m]:
            continue
        pre = f"output_layer.{hidden[m] + 1}."
        s1 = torch.zeros(W[m].shape[0], dtype=torch.float64, device=dev)
        s2 = torch.zeros_like(s1)
        for j0 in range(0, n, label_chunk):
            z = forward_chunk(j0, min(n, j0 + label_chunk), m)[4].double()
            s1 += z.sum(0)
            s2 += (z * z).sum(0)
        mean = s1 / rows
        var = (s2 / rows - mean * mean).clamp_min(0.0)
        fstat[m] = (mean.to(dt), (1.0 / torch.sqrt(var + eps)).to(dt), sd[pre + "weight"], sd[pre + "bias"])
        sd[pre + "running_mean"].mul_(1 - momentum).add_(mean.to(dt), alpha=momentum)
        sd[pre + "running_var"].mul_(1 - momentum).add_((var * (rows / max(rows - 1, 1))).to(dt), alpha=momentum)
        if pre + "num_batches_tracked" in sd:
            sd[pre + "num_batches_tracked"] += 1
    logits = torch.empty(b, n, dtype=dt, device=dev)
    for j0 in range(0, n, label_chunk):
        j1 = min(n, j0 + label_chunk)
        logits[:, j0:j1] = forward_chunk(j0, j1, nl)[4].reshape(b, j1 - j0)
    lg = logits.detach().clone().requires_grad_(True)
    y = multihots.to(dt)
    l = bce_loss(lg, y, **loss_kw) if loss == "BCE" else focal_loss(lg, y, **loss_kw)
    dl = torch.autograd.grad(l, lg)[0]  # [B, N_L]

    # ---- backward: S1 / S2 of each BatchNorm top down, then the accumulating pass
    S1 = [None] * nl
    S2 = [None] * nl

    def backward_chunk(j0, j1, down_to, final=False, acc=None):
        """du of layer `down_to` for labels [j0, j1) (and, in the final pass, every gradient contribution)."""
        x, zs, us, h_top, _ = forward_chunk(j0, j1, nl, keep=True)
        g = dl[:, j0:j1].reshape(-1, 1)  # rows i * nj + j, like the joint rows
        if final:
            acc["w_out"] += (g * h_top).sum(0, keepdim=True).double()
            acc["b_out"] += g.sum().double()
        du = (g * w_out) * (us[nl - 1] > 0)
        dz = None
        for m in range(nl - 1, -1, -1):
            if m < down_to:
                break
            if m == down_to and not final:
                return du, zs[m]
            if has_bn[m]:
                mean, invstd, gamma, _ = fstat[m]
                xhat = (zs[m] - mean) * invstd
                dz = (gamma * invstd) * (du - (S1[m] / rows).to(dt) - xhat * (S2[m] / rows).to(dt))
            else:
                dz = du
            if final:
                h_prev = F.relu(us[m - 1]) if m > 0 else x
                acc["W"][m] += (dz.t() @ h_prev).double()
                if not has_bn[m] and bias[m] is not None:
                    acc["bias"][m] += dz.sum(0).double()
            if m > 0:
                du = (dz @ W[m]) * (us[m - 1] > 0)
        dx = dz @ W[0]  # [rows, in_dim]
        nj = j1 - j0
        dx = dx.reshape(b, nj, -1)
        dP = dx[:, :, :d].sum(1).double()
        dL = dx[:, :, d:2 * d].sum(0).double()
        if fusion == "concatenation_diff":
            dP += dx[:, :, 2 * d:].sum(1).double()
            dL -= dx[:, :, 2 * d:].sum(0).double()
        if fusion == "concatenation_prod":
            dP += (dx[:, :, 2 * d:] * L_e[None, j0:j1, :]).sum(1).double()
            dL += (dx[:, :, 2 * d:] * P_e[:, None, :]).sum(0).double()
        acc["dP"] += dP
        acc["dL"][j0:j1] += dL
        return None, None

    for m in range(nl - 1, -1, -1):
        if not has_bn[m]:
            continue
        s1 = torch.zeros(W[m].shape[0], dtype=torch.float64, device=dev)
        s2 = torch.zeros_like(s1)
        mean, invstd, _, _ = fstat[m]
        for j0 in range(0, n, label_chunk):
            du, z = backward_chunk(j0, min(n, j0 + label_chunk), m)
            s1 += du.double().sum(0)
            s2 += (du * ((z - mean) * invstd)).double().sum(0)
        S1[m], S2[m] = s1, s2
    acc = {"W": [torch.zeros(w.shape, dtype=torch.float64, device=dev) for w in W],
           "bias": [torch.zeros(w.shape[0], dtype=torch.float64, device=dev) for w in W],
           "w_out": torch.zeros(w_out.shape, dtype=torch.float64, device=dev),
           "b_out": torch.zeros((), dtype=torch.float64, device=dev),
           "dP": torch.zeros(P_e.shape, dtype=torch.float64, device=dev),
           "dL": torch.zeros(L_e.shape, dtype=torch.float64, device=dev)}
    for j0 in range(0, n, label_chunk):
        backward_chunk(j0, min(n, j0 + label_chunk), 0, final=True, acc=acc)
    grads = {}
    for m, i in enumerate(hidden):
        grads[f"output_layer.{i}.weight"] = acc["W"][m].to(dt)
        if has_bn[m]:
            grads[f"output_layer.{i + 1}.weight"] = S2[m].to(dt)  # dgamma = sum du * xhat
            grads[f"output_layer.{i + 1}.bias"] = S1[m].to(dt)    # dbeta  = sum du
        elif bias[m] is not None:
            grads[f"output_layer.{i}.bias"] = acc["bias"][m].to(dt)
    grads[f"output_layer.{out_i}.weight"] = acc["w_out"].to(dt)
    if b_out is not None:
        grads[f"output_layer.{out_i}.bias"] = acc["b_out"].to(dt).reshape(b_out.shape)
    gl = torch.autograd.grad([P_e_g, L_e_g], [leaves[k] for k in head], grad_outputs=[acc["dP"].to(dt), acc["dL"].to(dt)])
    grads.update(dict(zip(head, gl)))
    return logits, l.detach(), grads


# ------------------------------------------------------------------------------------------------
# losses / metrics / optimiser step
# ------------------------------------------------------------------------------------------------
def bce_loss(logits: Tensor, target: Tensor, pos_weight: float = 1.0) -> Tensor:
    """losses.py:275-276."""
    return F.binary_cross_entropy_with_logits(logits, target,
                                              pos_weight=torch.tensor(pos_weight, dtype=logits.dtype, device=logits.device))


def focal_loss(logits: Tensor, target: Tensor, gamma: float = 2.0, alpha: float = -1.0,
               label_smoothing: float = 0.0) -> Tensor:
    """losses.py:190-213 (reduction='mean')."""
    if label_smoothing > 0:
        target = target * (1.0 - label_smoothing) + (1 - target) * label_smoothing
    bce = F.binary_cross_entropy_with_logits(logits, target, reduction="none")
    pt = torch.exp(-bce)
    loss = ((1 - pt) ** gamma) * bce
    if alpha >= 0:
        loss = (alpha * target + (1 - alpha) * (1 - target)) * loss
    return loss.mean()


def rgd_bce_loss(logits: Tensor, target: Tensor, temperature: float) -> Tensor:
    """losses.py:58-75 RGDBCE.  The reference passes the legacy argument `reduce="none"`, which torch reads as
    reduce=True, i.e. reduction 'mean': the re-weighting acts on the scalar mean loss, m * exp(min(m, T) / (T + 1))
    with the factor detached."""
    m = F.binary_cross_entropy_with_logits(logits, target)
    return m * torch.exp(torch.clamp(m.detach(), max=temperature) / (temperature + 1))


def supcon_loss(logits: Tensor, target: Tensor) -> Tensor:
    """losses.py:7-56 SupCon = one_way_supcon(dim=1): mean over rows of the mean log-softmax of the positives; a row
    without positives is 0/0 -> nan_to_num -> 0 in the value (its gradient stays NaN through autograd, as in the
    reference).  The temperature argument of the reference class is never used in its forward."""
    x = logits - logits.max(dim=1, keepdim=True)[0].detach()
    log_prob = x - torch.log(torch.exp(x).sum(1, keepdim=True))
    m = (target * log_prob).sum(1) / target.sum(1)
    return -torch.nan_to_num(m, 0).mean()


def batch_weights_v2(label_weights: Tensor, target: Tensor) -> Tensor:
    """losses.py:214-241: every element of row i weighs sum_j label_weights[j] * target[i, j]."""
    return (label_weights.float() * target).sum(dim=1, keepdim=True).expand_as(target)


def weighted_bce_loss(logits: Tensor, target: Tensor, label_weights: Tensor) -> Tensor:
    """losses.py:109-121 WeightedBCE."""
    return F.binary_cross_entropy_with_logits(logits, target, weight=batch_weights_v2(label_weights, target))


def cb_label_weights(label_counts: Tensor, beta: float = 0.9999) -> Tensor:
    """losses.py:86-99 CBLoss: (1 - beta) / (1 - beta^n_j), infinite effective number where it would be 0,
    normalised to sum to the number of classes."""
    eff = 1.0 - torch.pow(torch.tensor(beta), label_counts.float())
    eff = torch.where(eff == 0, torch.tensor(float("inf")), eff)
    w = (1.0 - beta) / eff
    return w / torch.sum(w) * len(label_counts)


def cb_loss(logits: Tensor, target: Tensor, label_counts: Tensor, beta: float = 0.9999) -> Tensor:
    """losses.py:78-106."""
    return weighted_bce_loss(logits, target, cb_label_weights(label_counts, beta))


def batch_weighted_bce_loss(logits: Tensor, target: Tensor, epsilon: float = 1e-10) -> Tensor:
    """losses.py:124-146 BatchWeightedBCE: positives and negatives of the batch weigh total/2 each."""
    num_pos = target.sum() + epsilon
    num_neg = target.numel() - num_pos + epsilon
    total = num_pos + num_neg
    w = target * ((1.0 / num_pos) * (total / 2.0)) + (1 - target) * ((1.0 / num_neg) * (total / 2.0))
    return F.binary_cross_entropy_with_logits(logits, target, weight=w)


def tp_fn_fp(probs: Tensor, labels: Tensor, threshold: float = 0.5):
    """ProtNoteTrainer.py:61-83."""
    preds = (probs >= threshold).float()
    return (preds * labels).sum(0), ((1 - preds) * labels).sum(0), (preds * (1 - labels)).sum(0)


def f1_per_label(tp, fn, fp):
    """ProtNoteTrainer.py:54-58."""
    pr = tp / (tp + fp + 1e-8)
    rc = tp / (tp + fn + 1e-8)
    return 2 * (pr * rc) / (pr + rc + 1e-8)


def f1_micro(tp, fn, fp):
    """ProtNoteTrainer.py:42-51."""
    return f1_per_label(tp.sum(), fn.sum(), fp.sum())


def trainable_names(sd: SD, train_sequence_encoder: bool = False, train_projection_head: bool = True):
    """ProtNoteTrainer.py:199-226: heads trainable; the encoder only with TRAIN_SEQUENCE_ENCODER (its classifier
    `sequence_encoder.output_layer` is never reached by get_embeddings, so it gets no gradient either way).
    TRAIN_PROJECTION_HEAD: False (:216-222) freezes every `output_layer.*` parameter; the reference's other test,
    name.startswith("W_p.weight") / ("W_l.weight"), is kept literally - no parameter is called that (they are
    W_p.0.weight, ...), so the projection heads go on training."""
    skip = ("running_mean", "running_var", "num_batches_tracked")
    out = []
    for k in sd:
        if k.startswith("label_encoder.") or k.endswith(skip):
            continue
        if k.startswith("sequence_encoder."):
            if not train_sequence_encoder or k.startswith("sequence_encoder.output_layer."):
                continue
        if (k.startswith("W_p.weight") or k.startswith("W_l.weight")) and not train_projection_head:
            continue
        if k.startswith("output_layer") and not train_projection_head:
            continue
        out.append(k)
    return out


def train_step(sd: SD, onehots: Tensor, lens: Tensor, label_embeddings: Tensor, multihots: Tensor, *,
               loss: str = "BCE", fusion: str = "concatenation", noise_alpha: float = 0.0,
               noise_u: Optional[Tensor] = None, label_token_counts: Optional[Tensor] = None,
               clip: Optional[float] = 1.0, lr: float = 3e-4, dilation_base: int = 3,
               adam_state: Optional[dict] = None, temperature: float = 0.07, apply_update: bool = True,
               train_sequence_encoder: bool = False, attention_mask: Optional[Tensor] = None,
               dropout_masks: Optional[dict] = None, train_projection_head: bool = True, optimizer: str = "Adam",
               weight_decay: float = 0.0, aux: Optional[dict] = None,
               **loss_kw) -> Tuple[Tensor, Tensor, Dict[str, Tensor], Tensor]:
    """Train-step body ProtNoteTrainer.py:728-755 (fp32; autocast/GradScaler are no-ops on CPU) with the optimiser
    _set_optimizer built (:230-243): Adam(lr) | AdamW(lr, weight_decay) | SGD(lr, weight_decay) at torch's defaults.

    Updates `sd` in place (params by the optimiser, BN buffers by the train-mode forward).
    Returns (logits, loss, grads, total_grad_norm)."""
    names = trainable_names(sd, train_sequence_encoder, train_projection_head)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    logits = protnote_forward(work, onehots, lens, label_embeddings, fusion=fusion, training=True,
                              noise_alpha=noise_alpha, noise_u=noise_u, temperature=temperature,
                              label_token_counts=label_token_counts, dilation_base=dilation_base,
                              train_sequence_encoder=train_sequence_encoder, attention_mask=attention_mask,
                              dropout_masks=dropout_masks, aux=aux)
    y = multihots.float()
    l = bce_loss(logits, y, **loss_kw) if loss == "BCE" else focal_loss(logits, y, **loss_kw)
    grads_t = torch.autograd.grad(l, [leaves[k] for k in names], allow_unused=True)
    grads = {k: g for k, g in zip(names, grads_t) if g is not None}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    if apply_update:
        coef = 1.0
        if clip is not None:  # torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm/(norm+1e-6), max=1)
            coef = min(float(clip) / (float(total) + 1e-6), 1.0)
        st = adam_state if adam_state is not None else {}
        st["step"] = st.get("step", 0) + 1
        t = st["step"]
        b1, b2, eps = 0.9, 0.999, 1e-8
        for k, g in grads.items():
            g = g * coef
            if optimizer == "SGD":  # torch.optim.SGD, momentum 0: p -= lr (g + wd p)
                sd[k] = sd[k] - lr * (g + weight_decay * sd[k])
                continue
            if optimizer == "AdamW":  # decoupled decay before the Adam update
                sd[k] = sd[k] * (1 - lr * weight_decay)
            m = st.setdefault("m/" + k, torch.zeros_like(g))
            v = st.setdefault("v/" + k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            sd[k] = sd[k] - (lr / (1 - b1 ** t)) * (m / denom)
    return logits.detach(), l.detach(), grads, total


def as_torch_sd(npz, prefix: str) -> SD:
    """Load 'prefix/<state-dict key>' arrays of a golden .npz as a torch state dict (copies)."""
    import numpy as np

    out = {}
    for k in npz.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(npz[k]))
    return out
