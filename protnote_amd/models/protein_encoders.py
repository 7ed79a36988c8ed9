"""ProteInfer encoder on MI355X - drop-in twin of protnote/models/protein_encoders.py (reference :8-153).

Same constructor, methods, and state_dict keys as the reference classes; torch.nn modules are used only as
parameter containers (device memory + checkpoint key layout).  All arithmetic - masked dilated
convolutions as f32-MFMA implicit GEMMs with the BatchNorm/ReLU/padding-mask fused into operand load and
epilogue, and the masked mean-pool - runs in libprotnote_hip.so (pn_encoder_fwd)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L


def _pack_conv_weight(conv: torch.nn.Conv1d) -> torch.Tensor:
    """torch [Cout][Cin][k] -> the kernels' [Cout][k][ld4(Cin)] (pn_pack_conv_weight)."""
    w = conv.weight
    cout, cin, k = w.shape
    packed = torch.empty(cout, k, _ld4(cin), dtype=torch.float32, device=w.device)
    L.check(L.lib().pn_pack_conv_weight(L.ptr(w.detach().contiguous()), L.ptr(packed), cout, cin, k, L.stream_ptr()))
    return packed


def _conv_geometry(conv: torch.nn.Conv1d, who: str):
    k, dil = int(conv.kernel_size[0]), int(conv.dilation[0])
    if conv.padding != "same" or int(conv.stride[0]) != 1 or conv.groups != 1 or k % 2 != 1:
        raise ValueError(f"{who}: the kernels implement Conv1d(padding='same', stride=1, groups=1, odd kernel_size) - what "
                         "the reference's ProteInfer builds (protein_encoders.py:39-59,84-91)")
    return k, dil


def _ncl_input(x, sequence_lengths, channels, who):
    L.require_hip(x, sequence_lengths)
    if x.dim() != 3 or x.shape[1] != channels:
        raise ValueError(f"{who}: expected [B, {channels}, L] input, got {tuple(x.shape)}")
    x = x.detach().contiguous().float()
    lens = sequence_lengths.detach().to(device=x.device, dtype=torch.int64).contiguous()
    return x, lens


class MaskedConv1D(torch.nn.Conv1d):
    """Reference MaskedConv1D (protein_encoders.py:8-17).  Inside ProteInfer the convolutions run fused and channels-last
    (pn_encoder_fwd); called on its own, `forward` is the reference's mask -> Conv1d(padding='same') -> mask on [B, C, L]
    tensors through pn_masked_conv1d_fwd (inference only: no autograd through a stand-alone call)."""

    def forward(self, x, sequence_lengths):
        k, dil = _conv_geometry(self, "MaskedConv1D")
        x, lens = _ncl_input(x, sequence_lengths, self.in_channels, "MaskedConv1D")
        B, _, Lmax = x.shape
        lib = L.lib()
        packed = _pack_conv_weight(self)
        ws = L.workspace(lib.pn_masked_conv1d_ws_bytes(B, Lmax, self.in_channels, self.out_channels), x.device, "conv1d")
        out = torch.empty(B, self.out_channels, Lmax, dtype=torch.float32, device=x.device)
        L.check(lib.pn_masked_conv1d_fwd(L.ptr(x), L.ptr(lens), L.ptr(packed), L.ptr(self.bias.detach()) if self.bias is not None
                                         else None, B, self.in_channels, self.out_channels, Lmax, k, dil, L.ptr(out), L.ptr(ws),
                                         ws.numel(), L.stream_ptr()))
        return out


class Residual(torch.nn.Module):
    """Reference Residual (protein_encoders.py:23-67).  Inside ProteInfer the block runs fused (pn_encoder_fwd); called on its
    own, `forward` is the reference's bn1 -> ReLU -> masked_conv1 -> bn2 -> ReLU -> masked_conv2 -> + x on [B, C, L] tensors
    through pn_residual_fwd - including what distinguishes a stand-alone call: bn1 normalises the RAW input (train-mode
    statistics run over the pad positions too) and the input is added back unmasked.  Inference / buffer updates only: no
    autograd through a stand-alone call."""

    def __init__(self, input_channels: int, kernel_size: int, dilation: int, bottleneck_factor: float,
                 activation=torch.nn.ReLU):
        super().__init__()
        if activation is not torch.nn.ReLU:
            raise ValueError("protnote_amd Residual implements the ReLU activation only")
        bottleneck = int(np.floor(input_channels * bottleneck_factor))
        self.bn_activation_1 = torch.nn.Sequential(
            torch.nn.BatchNorm1d(input_channels, eps=0.001, momentum=0.01), activation())
        self.masked_conv1 = MaskedConv1D(input_channels, bottleneck, kernel_size=kernel_size, stride=1,
                                         padding="same", dilation=dilation)
        self.bn_activation_2 = torch.nn.Sequential(
            torch.nn.BatchNorm1d(bottleneck, eps=0.001, momentum=0.01), activation())
        self.masked_conv2 = MaskedConv1D(bottleneck, input_channels, kernel_size=1, stride=1, padding="same",
                                         dilation=1)

    def forward(self, x, sequence_lengths):
        k, dil = _conv_geometry(self.masked_conv1, "Residual")
        C_, Cb = self.masked_conv1.in_channels, self.masked_conv1.out_channels
        x, lens = _ncl_input(x, sequence_lengths, C_, "Residual")
        B, _, Lmax = x.shape
        bn1, bn2 = self.bn_activation_1[0], self.bn_activation_2[0]
        if self.training and B * Lmax == 1:  # torch's BatchNorm1d refuses batch statistics over one value per channel
            raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([1, {C_}, 1])")
        lib = L.lib()
        blk = L.pn_res_block()
        pa, pb = _pack_conv_weight(self.masked_conv1), _pack_conv_weight(self.masked_conv2)
        blk.bn1 = L.pn_bn(bn1.weight.data_ptr(), bn1.bias.data_ptr(), bn1.running_mean.data_ptr(), bn1.running_var.data_ptr())
        blk.conv_a_w, blk.conv_a_b = pa.data_ptr(), self.masked_conv1.bias.data_ptr()
        blk.bn2 = L.pn_bn(bn2.weight.data_ptr(), bn2.bias.data_ptr(), bn2.running_mean.data_ptr(), bn2.running_var.data_ptr())
        blk.conv_b_w, blk.conv_b_b = pb.data_ptr(), self.masked_conv2.bias.data_ptr()
        ws = L.workspace(lib.pn_residual_ws_bytes(B, Lmax, C_, Cb), x.device, "residual")
        out = torch.empty_like(x)
        L.check(lib.pn_residual_fwd(C.byref(blk), C_, Cb, k, dil, L.ptr(x), L.ptr(lens), B, Lmax, L.ptr(out),
                                    1 if self.training else 0, L.ptr(ws), ws.numel(), L.stream_ptr()))
        if self.training:
            torch._foreach_add_([bn1.num_batches_tracked, bn2.num_batches_tracked], 1)
        del pa, pb
        return out


def _ld4(c):
    return (c + 3) & ~3


class ResidueIds:
    """A batch of sequences as the device-side collator holds it (protnote_amd.data.collators.collate_to_device with
    residue_ids=True): the residue indices back to back (uint8, on the device) + offsets [B + 1] (int64) - in place of the
    reference's f32 one-hots [B, A, Lmax] (collators.py:123-133).  ProteInfer.get_embeddings / ProtNote.forward accept it
    where they accept `sequence_onehots`: the encoder then starts from the ids (pn_encoder_fwd_ids) instead of writing the
    one-hots and re-deriving the ids from them.  `.to_onehots()` gives the reference's tensor."""

    def __init__(self, flat, offsets, alphabet_size: int, max_length: int):
        self.flat, self.offsets = flat, offsets
        self.alphabet_size, self.max_length = int(alphabet_size), int(max_length)

    @property
    def shape(self):
        return (int(self.offsets.numel()) - 1, self.alphabet_size, self.max_length)

    @property
    def device(self):
        return self.flat.device

    @property
    def is_cuda(self):
        return self.flat.is_cuda

    def dim(self):
        return 3

    def to(self, *args, **kwargs):
        return ResidueIds(self.flat.to(*args, **kwargs), self.offsets.to(*args, **kwargs), self.alphabet_size, self.max_length)

    def lengths(self):
        return (self.offsets[1:] - self.offsets[:-1]).clamp(max=self.max_length)

    def to_onehots(self):
        """The reference collator's zero-padded one-hots [B, A, Lmax] f32 and lengths [B] i64 (pn_onehot_batch)."""
        L.require_hip(self.flat, self.offsets)
        B, A, Lmax = self.shape
        onehots = torch.empty(B, A, Lmax, dtype=torch.float32, device=self.flat.device)
        lengths = torch.empty(B, dtype=torch.int64, device=self.flat.device)
        L.check(L.lib().pn_onehot_batch(L.ptr(self.flat), L.ptr(self.offsets), B, A, Lmax, L.ptr(onehots), L.ptr(lengths),
                                        L.stream_ptr()))
        return onehots, lengths


class ProteInfer(torch.nn.Module):
    def __init__(self, num_labels: int, input_channels: int, output_channels: int, kernel_size: int, activation,
                 dilation_base: int, num_resnet_blocks: int, bottleneck_factor: float):
        super().__init__()
        if activation is not torch.nn.ReLU:
            raise ValueError("protnote_amd ProteInfer implements the ReLU activation only")
        if num_resnet_blocks > L.PN_MAX_BLOCKS:
            raise ValueError(f"at most {L.PN_MAX_BLOCKS} residual blocks are supported")
        if kernel_size % 2 != 1:
            raise ValueError("kernel_size must be odd (padding='same')")
        self.conv1 = MaskedConv1D(input_channels, output_channels, kernel_size=kernel_size, stride=1,
                                  padding="same", dilation=1)
        self.resnet_blocks = torch.nn.ModuleList(
            Residual(output_channels, kernel_size, dilation_base ** i, bottleneck_factor, activation)
            for i in range(num_resnet_blocks))
        self.output_layer = torch.nn.Linear(output_channels, num_labels)
        self._dims = dict(Cin=input_channels, C=output_channels,
                          Cb=int(np.floor(output_channels * bottleneck_factor)), ksize=kernel_size,
                          nblocks=num_resnet_blocks, dil_base=dilation_base)
        self._packed = {}  # name -> (version, data_ptr, packed tensor)

    # ---- weight packing: torch [Cout][Cin][k] -> [Cout][k][ld4(Cin)] (pn_pack_conv_weight) ----
    def _pack(self, name: str, conv: torch.nn.Conv1d) -> torch.Tensor:
        w = conv.weight
        # (inference tensors carry no version counter: repacked on every call, 11 small launches)
        key = None if w.is_inference() else (w._version, w.data_ptr(), w.device)
        hit = self._packed.get(name)
        # a trainable weight may be updated through raw pointers (FusedClipAdam) without a version bump: repack
        if key is not None and hit is not None and hit[0] == key and not w.requires_grad:
            return hit[1]
        packed = _pack_conv_weight(conv)
        self._packed[name] = (key, packed)
        return packed

    @staticmethod
    def _bn_struct(bn: torch.nn.BatchNorm1d) -> L.pn_bn:
        return L.pn_bn(bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                       bn.running_var.data_ptr())

    def _descriptor(self):
        d = self._dims
        enc = L.pn_encoder()
        enc.Cin, enc.C, enc.Cb = d["Cin"], d["C"], d["Cb"]
        enc.ksize, enc.nblocks, enc.dil_base = d["ksize"], d["nblocks"], d["dil_base"]
        # per-call arithmetic (None = the process default now); ProtNote.math_mode's setter writes it through
        enc.math_mode = L.math_field(getattr(self, "math_mode", None))
        keep = [self._pack("conv1", self.conv1)]
        enc.conv1_w = keep[-1].data_ptr()
        enc.conv1_b = self.conv1.bias.data_ptr()
        for i, blk in enumerate(self.resnet_blocks):
            b = enc.blk[i]
            b.bn1 = self._bn_struct(blk.bn_activation_1[0])
            keep.append(self._pack(f"a{i}", blk.masked_conv1))
            b.conv_a_w = keep[-1].data_ptr()
            b.conv_a_b = blk.masked_conv1.bias.data_ptr()
            b.bn2 = self._bn_struct(blk.bn_activation_2[0])
            keep.append(self._pack(f"b{i}", blk.masked_conv2))
            b.conv_b_w = keep[-1].data_ptr()
            b.conv_b_b = blk.masked_conv2.bias.data_ptr()
        return enc, keep

    def trunk_parameters(self):
        """Parameters reached by get_embeddings, in the order _EncoderTrainFn returns their gradients."""
        ps = [self.conv1.weight, self.conv1.bias]
        for blk in self.resnet_blocks:
            bn1, bn2 = blk.bn_activation_1[0], blk.bn_activation_2[0]
            ps += [bn1.weight, bn1.bias, blk.masked_conv1.weight, blk.masked_conv1.bias, bn2.weight, bn2.bias,
                   blk.masked_conv2.weight, blk.masked_conv2.bias]
        return ps

    def get_embeddings(self, x, sequence_lengths):
        """[B, Cin, L] f32 one-hots + [B] lengths -> [B, C] masked mean-pooled features
        (reference protein_encoders.py:109-118).  In train mode BatchNorm uses batch statistics and
        updates its running buffers exactly like the reference's "frozen" encoder does.
        With gradients enabled and trainable parameters (TRAIN_SEQUENCE_ENCODER: True) the call is differentiable:
        pn_encoder_fwd_train keeps the activations, pn_encoder_bwd returns every parameter gradient."""
        if self.training and x.dim() == 3 and x.shape[0] * x.shape[2] == 1 and len(self.resnet_blocks) > 0:
            # torch's BatchNorm1d refuses batch statistics over one value per channel (a single residue in the whole batch):
            # the reference raises here, so does the twin
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([1, {self._dims['C']}, 1])")
        from_ids = isinstance(x, ResidueIds)
        differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in self.trunk_parameters())
        if from_ids and (differentiable or not self._ids_path_ok(x)):
            x, sequence_lengths = x.to_onehots()  # trainable encoder / a conv1 the gather form does not cover: the one-hot route
            from_ids = False
        if differentiable:
            # (eval mode: BatchNorm normalises with its running statistics, which the backward treats as constants -
            #  pn_encoder.bn_use_running)
            return _EncoderTrainFn.apply(self, x, sequence_lengths, *self.trunk_parameters())
        if from_ids:
            return self._embeddings_from_ids(x)
        L.require_hip(x, sequence_lengths)
        if x.dim() != 3 or x.shape[1] != self._dims["Cin"]:
            raise ValueError(f"expected [B, {self._dims['Cin']}, L] input, got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        lens = sequence_lengths.detach().to(device=x.device, dtype=torch.int64).contiguous()
        B, _, Lmax = x.shape
        enc, keep = self._descriptor()
        lib = L.lib()
        nbytes = lib.pn_encoder_ws_bytes(C.byref(enc), B, Lmax)
        ws = L.workspace(nbytes, x.device, "enc")
        emb = torch.empty(B, self._dims["C"], dtype=torch.float32, device=x.device)
        training = 1 if self.training else 0
        L.check(lib.pn_encoder_fwd(C.byref(enc), L.ptr(x), L.ptr(lens), B, Lmax, L.ptr(emb), emb.shape[1],
                                   training, L.ptr(ws), ws.numel(), L.stream_ptr()))
        if training:
            self._bump_batches_tracked()
        del keep
        return emb

    def _ids_path_ok(self, ids: "ResidueIds") -> bool:
        d = self._dims
        return ids.alphabet_size == d["Cin"] and d["ksize"] == 9 and d["ksize"] * (d["Cin"] + 1) <= 255

    def _embeddings_from_ids(self, ids: "ResidueIds"):
        """get_embeddings on a ResidueIds batch: pn_encoder_fwd_ids (bit-identical to the one-hot route)."""
        L.require_hip(ids.flat, ids.offsets)
        B, _, Lmax = ids.shape
        enc, keep = self._descriptor()
        lib = L.lib()
        ws = L.workspace(lib.pn_encoder_ws_bytes(C.byref(enc), B, Lmax), ids.device, "enc")
        emb = torch.empty(B, self._dims["C"], dtype=torch.float32, device=ids.device)
        training = 1 if self.training else 0
        L.check(lib.pn_encoder_fwd_ids(C.byref(enc), L.ptr(ids.flat), L.ptr(ids.offsets), B, Lmax, L.ptr(emb), emb.shape[1],
                                       training, L.ptr(ws), ws.numel(), L.stream_ptr()))
        if training:
            self._bump_batches_tracked()
        del keep
        return emb

    def _bump_batches_tracked(self):
        tracked = [bn.num_batches_tracked for blk in self.resnet_blocks
                   for bn in (blk.bn_activation_1[0], blk.bn_activation_2[0])]
        if tracked:  # num_resnet_blocks == 0: no BatchNorm at all
            torch._foreach_add_(tracked, 1)  # one multi-tensor launch

    def forward(self, x, sequence_lengths):
        """Reference protein_encoders.py:120-123: Linear(C -> num_labels) on the pooled features."""
        feats = self.get_embeddings(x, sequence_lengths)
        w, b = self.output_layer.weight, self.output_layer.bias
        if feats.shape[1] % 4 != 0:
            raise ValueError("output_channels must be a multiple of 4")
        out = torch.empty(feats.shape[0], w.shape[0], dtype=torch.float32, device=feats.device)
        L.check(L.lib().pn_gemm_nt(L.ptr(feats), feats.shape[1], L.ptr(w.detach()), w.shape[1], L.ptr(out),
                                   out.shape[1], feats.shape[0], w.shape[0], w.shape[1], L.ptr(b.detach()),
                                   None, None, None, None, -1, None, 0, L.stream_ptr()))
        return out

    @classmethod
    def from_pretrained(cls, weights_path: str, num_labels: int, input_channels: int, output_channels: int,
                        kernel_size: int, activation, dilation_base: int, num_resnet_blocks: int,
                        bottleneck_factor: float):
        """Reference protein_encoders.py:125-153 + utils/proteinfer.py:7-41 (TF-variable pickle)."""
        from ..utils.proteinfer import transfer_tf_weights_to_torch

        model = cls(num_labels, input_channels, output_channels, kernel_size, activation, dilation_base,
                    num_resnet_blocks, bottleneck_factor)
        transfer_tf_weights_to_torch(model, weights_path)
        return model


class _EncoderTrainFn(torch.autograd.Function):
    """Differentiable ProteInfer.get_embeddings (TRAIN_SEQUENCE_ENCODER: True, reference ProtNote.py:248-256)."""

    @staticmethod
    def forward(ctx, enc_mod, x, lens, *params):
        L.require_hip(x, lens)
        x = x.detach().contiguous().float()
        lens = lens.detach().to(device=x.device, dtype=torch.int64).contiguous()
        B, _, Lmax = x.shape
        enc, keep = enc_mod._descriptor()
        ctx.bn_running = enc.bn_use_running = 0 if enc_mod.training else 1
        ctx.math_field = enc.math_mode  # the backward runs in the arithmetic its forward ran in
        lib = L.lib()
        save = torch.empty(lib.pn_encoder_train_save_bytes(C.byref(enc), B, Lmax), dtype=torch.uint8, device=x.device)
        ws = L.workspace(lib.pn_encoder_ws_bytes(C.byref(enc), B, Lmax), x.device, "enc")
        emb = torch.empty(B, enc_mod._dims["C"], dtype=torch.float32, device=x.device)
        L.check(lib.pn_encoder_fwd_train(C.byref(enc), L.ptr(x), L.ptr(lens), B, Lmax, L.ptr(emb), emb.shape[1],
                                         L.ptr(save), save.numel(), L.ptr(ws), ws.numel(), L.stream_ptr()))
        if enc_mod.training:
            enc_mod._bump_batches_tracked()
        ctx.enc_mod, ctx.save, ctx.shape, ctx.params = enc_mod, save, (B, Lmax), params
        del keep
        return emb

    @staticmethod
    def backward(ctx, demb):
        enc_mod, (B, Lmax) = ctx.enc_mod, ctx.shape
        enc, keep = enc_mod._descriptor()
        enc.bn_use_running = ctx.bn_running
        enc.math_mode = ctx.math_field
        lib = L.lib()
        demb = demb.contiguous().float()
        grads = [torch.empty_like(p, memory_format=torch.contiguous_format) for p in ctx.params]
        gr = L.pn_encoder_grads()
        gr.conv1_w, gr.conv1_b = grads[0].data_ptr(), grads[1].data_ptr()
        names = ("bn1_w", "bn1_b", "conv_a_w", "conv_a_b", "bn2_w", "bn2_b", "conv_b_w", "conv_b_b")
        for i in range(enc_mod._dims["nblocks"]):
            for j, n in enumerate(names):
                setattr(gr.blk[i], n, grads[2 + 8 * i + j].data_ptr())
        ws = L.workspace(lib.pn_encoder_bwd_ws_bytes(C.byref(enc), B, Lmax), demb.device, "encbwd")
        L.check(lib.pn_encoder_bwd(C.byref(enc), B, Lmax, L.ptr(demb), demb.shape[1], C.byref(gr), L.ptr(ctx.save),
                                   ctx.save.numel(), L.ptr(ws), ws.numel(), L.stream_ptr()))
        del keep
        ctx.save = None
        outs = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[3:])]
        return (None, None, None, *outs)
