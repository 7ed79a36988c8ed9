"""ProteInfer encoder on MI355X - drop-in twin of protnote/models/protein_encoders.py (reference :8-153).

Same constructor, methods, and state_dict keys as the reference classes; torch.nn modules are used only as
parameter containers (device memory + checkpoint key layout).  All arithmetic - masked dilated
convolutions as f32-MFMA implicit GEMMs with the BatchNorm/ReLU/padding-mask fused into operand load and
epilogue, and the masked mean-pool - runs in libprotnote_hip.so (pn_encoder_fwd)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L


class MaskedConv1D(torch.nn.Conv1d):
    """Parameter container for reference MaskedConv1D (protein_encoders.py:8-17)."""

    def forward(self, x, sequence_lengths):  # pragma: no cover - never called piecewise
        raise RuntimeError("MaskedConv1D runs fused inside pn_encoder_fwd; call ProteInfer.get_embeddings")


class Residual(torch.nn.Module):
    """Parameter container for reference Residual (protein_encoders.py:23-67)."""

    def __init__(self, input_channels: int, kernel_size: int, dilation: int, bottleneck_factor: float,
                 activation=torch.nn.ReLU):
        super().__init__()
        bottleneck = int(np.floor(input_channels * bottleneck_factor))
        self.bn_activation_1 = torch.nn.Sequential(
            torch.nn.BatchNorm1d(input_channels, eps=0.001, momentum=0.01), activation())
        self.masked_conv1 = MaskedConv1D(input_channels, bottleneck, kernel_size=kernel_size, stride=1,
                                         padding="same", dilation=dilation)
        self.bn_activation_2 = torch.nn.Sequential(
            torch.nn.BatchNorm1d(bottleneck, eps=0.001, momentum=0.01), activation())
        self.masked_conv2 = MaskedConv1D(bottleneck, input_channels, kernel_size=1, stride=1, padding="same",
                                         dilation=1)

    def forward(self, x, sequence_lengths):  # pragma: no cover
        raise RuntimeError("Residual runs fused inside pn_encoder_fwd; call ProteInfer.get_embeddings")


def _ld4(c):
    return (c + 3) & ~3


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
        cout, cin, k = w.shape
        packed = torch.empty(cout, k, _ld4(cin), dtype=torch.float32, device=w.device)
        L.check(L.lib().pn_pack_conv_weight(L.ptr(w.detach().contiguous()), L.ptr(packed), cout, cin, k,
                                            L.stream_ptr()))
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
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.trunk_parameters()):
            # (eval mode: BatchNorm normalises with its running statistics, which the backward treats as constants -
            #  pn_encoder.bn_use_running)
            return _EncoderTrainFn.apply(self, x, sequence_lengths, *self.trunk_parameters())
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
