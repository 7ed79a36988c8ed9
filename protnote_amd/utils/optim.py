"""Fused gradient clipping + Adam on MI355X (the optimiser half of the reference train step,
ProtNoteTrainer.py:745-755: clip_grad_norm_(max_norm) then Adam(lr).step()).

All trainable parameters, their gradients and both Adam moments live in four flat f32 buffers; each
nn.Parameter (and its .grad) is a view into them, so the whole step is two kernels (sum of squares, then
clip + Adam) and the data-parallel gradient exchange is ONE RCCL all-reduce over the flat gradient."""
import torch

from .. import _lib as L


class FusedClipAdam:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=1.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        L.require_hip(*self.params)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep every view 16-byte aligned
        n = sum(sizes)
        self.flat_w = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p, sz in zip(self.params, sizes):
                view = self.flat_w[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[off:off + p.numel()].view_as(p)
                off += sz
        self.step_count = 0
        self.last_grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()
        off = 0
        for p in self.params:  # re-attach views if something replaced them
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + p.numel()].view_as(p)
            off += (p.numel() + 3) // 4 * 4

    def step(self):
        self.step_count += 1
        ws = L.workspace(256, self.flat_w.device, "adam")
        max_norm = -1.0 if self.max_norm is None else float(self.max_norm)
        L.check(L.lib().pn_clip_adam_step(L.ptr(self.flat_w), L.ptr(self.flat_g), L.ptr(self.flat_m),
                                          L.ptr(self.flat_v), self.flat_w.numel(), max_norm, float(self.lr),
                                          float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                          float(self.weight_decay), self.step_count, L.ptr(self.last_grad_norm),
                                          L.ptr(ws), ws.numel(), L.stream_ptr()))

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.flat_m, "exp_avg_sq": self.flat_v}
