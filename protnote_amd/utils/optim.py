"""Fused gradient clipping + Adam on MI355X (the optimiser half of the reference train step,
ProtNoteTrainer.py:745-755: clip_grad_norm_(max_norm) then Adam(lr).step()).

All trainable parameters, their gradients and both Adam moments live in four flat f32 buffers; each
nn.Parameter (and its .grad) is a view into them, so the whole step is two kernels (sum of squares, then
clip + Adam) and the data-parallel gradient exchange is ONE RCCL all-reduce over the flat gradient."""
import torch

from .. import _lib as L


_weights_generation = 0


def weights_generation() -> int:
    """Bumped whenever a fused optimiser rewrites parameters through raw device pointers (step / repack /
    load_state_dict): tensor version counters do not see those writes, caches of weight-derived tensors key on this."""
    return _weights_generation


def _bump_generation():
    global _weights_generation
    _weights_generation += 1


class FusedClipAdam:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=1.0,
                 param_ids=None, n_param_ids=None, _moments=2):
        """`param_ids[i]` = the id parameter i carries in state_dict() / load_state_dict() and `n_param_ids` the length of
        the id space: the position of the parameter in the list the REFERENCE builds its torch.optim.Adam from
        (model.named_parameters() filtered by requires_grad, ProtNoteTrainer.py:199-231), which may hold parameters this
        optimiser does not own (e.g. the ProteInfer classifier under TRAIN_SEQUENCE_ENCODER: trainable there, never
        reached by get_embeddings, so torch keeps no state for it).  Default: position in `params`."""
        params = list(params)
        keep = [i for i, p in enumerate(params) if p.requires_grad]
        self.params = [params[i] for i in keep]
        if not self.params:
            raise ValueError("no trainable parameters")
        if param_ids is not None:
            if len(param_ids) != len(params):
                raise ValueError("param_ids must have one entry per parameter")
            self.param_ids = [int(param_ids[i]) for i in keep]
            self.n_param_ids = int(n_param_ids) if n_param_ids is not None else max(self.param_ids) + 1
            if len(set(self.param_ids)) != len(self.param_ids) or max(self.param_ids) >= self.n_param_ids:
                raise ValueError("param_ids must be distinct and below n_param_ids")
        else:
            self.param_ids = list(range(len(self.params)))
            self.n_param_ids = len(self.params)
        dev = self.params[0].device
        L.require_hip(*self.params)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep every view 16-byte aligned
        n = sum(sizes)
        self.flat_w = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        # `_moments`: how many per-parameter state blocks the update rule keeps (Adam 2; SGD 1 with momentum, else 0)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev) if _moments >= 1 else None
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev) if _moments >= 2 else None
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

    def state_buffers(self):
        """The flat blocks a replica must share with rank 0 at start / resume (distributed.sync_initial_state): the
        weights and whatever moment blocks this update rule keeps (SGD without momentum keeps none)."""
        return [t for t in (self.flat_w, self.flat_m, self.flat_v) if t is not None]

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()
        for p, off in self._offsets():  # re-attach views if something replaced them
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + p.numel()].view_as(p)

    def _offsets(self):
        off = 0
        for p in self.params:
            yield p, off
            off += (p.numel() + 3) // 4 * 4

    def step(self):
        for p, off in self._offsets():
            # model.to() / .float() / load_state_dict(assign=True) replace p.data: the kernel would then update a
            # buffer the model no longer reads.  Fail loudly instead of training a detached copy.
            if p.data_ptr() != self.flat_w.data_ptr() + 4 * off:
                raise RuntimeError("FusedClipAdam: a parameter no longer aliases the flat weight buffer (the model "
                                   "was moved / cast / re-assigned after the optimiser was built); rebuild the "
                                   "optimiser, or call repack()")
        self.step_count += 1
        _bump_generation()
        ws = L.workspace(L.PN_ADAM_WS_BYTES, self.flat_w.device, "adam")
        max_norm = -1.0 if self.max_norm is None else float(self.max_norm)
        L.check(L.lib().pn_clip_adam_step(L.ptr(self.flat_w), L.ptr(self.flat_g), L.ptr(self.flat_m),
                                          L.ptr(self.flat_v), self.flat_w.numel(), max_norm, float(self.lr),
                                          float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                          float(self.weight_decay), self.step_count, L.ptr(self.last_grad_norm),
                                          L.ptr(ws), ws.numel(), L.stream_ptr()))

    def repack(self):
        """Re-adopt the parameters' current values (after the model was moved / re-assigned) as views of flat_w."""
        _bump_generation()
        with torch.no_grad():
            for p, off in self._offsets():
                view = self.flat_w[off:off + p.numel()].view_as(p)
                if p.data_ptr() != view.data_ptr():
                    view.copy_(p.data)
                    p.data = view
        self.zero_grad()

    # ---- checkpoint interchange with torch.optim.Adam / AdamW (reference utils/models.py:304-321,366-367) ----
    def state_dict(self):
        """The layout torch.optim.Adam.state_dict() produces for the reference's parameter list (ids = `param_ids`, see
        __init__), so `optimizer_state_dict` of a checkpoint moves both ways between this optimiser and the reference's
        Adam."""
        state = {}
        if self.step_count > 0:
            for i, (p, off) in zip(self.param_ids, self._offsets()):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.flat_m[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.flat_v[off:off + n].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(self.n_param_ids))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:  # round-1 flat layout {step, exp_avg, exp_avg_sq}
            self.step_count = int(sd["step"])
            self.flat_m.copy_(sd["exp_avg"])
            self.flat_v.copy_(sd["exp_avg_sq"])
            return
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) == self.n_param_ids:
            slots = self.param_ids  # the reference's id space (position in its requires_grad-filtered parameter list)
        elif len(ids) == len(self.params):
            # rounds 1-2 of this framework numbered the parameters by their position in ITS OWN list; the two spaces only
            # differ under TRAIN_SEQUENCE_ENCODER (the reference's list also holds the never-trained ProteInfer classifier)
            slots = list(range(len(self.params)))
        else:
            raise ValueError(f"optimizer state has {len(ids)} parameters; expected {self.n_param_ids} (the reference's "
                             f"parameter list) or {len(self.params)} (this optimiser's own, the pre-round-3 layout)")
        _bump_generation()
        g0 = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g0["lr"], tuple(g0["betas"]), g0["eps"]
        self.weight_decay = g0.get("weight_decay", 0.0)
        steps = set()
        with torch.no_grad():
            self.flat_m.zero_()
            self.flat_v.zero_()
            for k, (p, off) in zip(slots, self._offsets()):
                pid = ids[k]
                st = sd["state"].get(pid)
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {pid}: shape {tuple(st['exp_avg'].shape)} vs {tuple(p.shape)}")
                n = p.numel()
                self.flat_m[off:off + n].view_as(p).copy_(st["exp_avg"])
                self.flat_v[off:off + n].view_as(p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused step keeps one count")
        self.step_count = steps.pop() if steps else 0


class FusedClipSGD(FusedClipAdam):
    """OPTIMIZER: SGD - clip_grad_norm_ + torch.optim.SGD(lr, weight_decay) (reference ProtNoteTrainer.py:238-243, which
    leaves momentum / dampening / nesterov at torch's defaults 0 / 0 / False) on the same flat buffers: two kernels per
    step, one RCCL all-reduce of the flat gradient.  `momentum` != 0 follows torch (velocity in flat_m)."""

    def __init__(self, params, lr=3e-4, momentum=0.0, weight_decay=0.0, max_norm=1.0, param_ids=None, n_param_ids=None):
        super().__init__(params, lr=lr, weight_decay=weight_decay, max_norm=max_norm, param_ids=param_ids,
                         n_param_ids=n_param_ids, _moments=1 if float(momentum) != 0.0 else 0)  # no second moment
        self.momentum = float(momentum)

    def step(self):
        for p, off in self._offsets():
            if p.data_ptr() != self.flat_w.data_ptr() + 4 * off:
                raise RuntimeError("FusedClipSGD: a parameter no longer aliases the flat weight buffer (the model was "
                                   "moved / cast / re-assigned after the optimiser was built); rebuild the optimiser, "
                                   "or call repack()")
        self.step_count += 1
        _bump_generation()
        ws = L.workspace(L.PN_ADAM_WS_BYTES, self.flat_w.device, "adam")
        max_norm = -1.0 if self.max_norm is None else float(self.max_norm)
        L.check(L.lib().pn_clip_sgd_step(L.ptr(self.flat_w), L.ptr(self.flat_g), L.ptr(self.flat_m),
                                         self.flat_w.numel(), max_norm, float(self.lr), self.momentum,
                                         float(self.weight_decay), self.step_count, L.ptr(self.last_grad_norm),
                                         L.ptr(ws), ws.numel(), L.stream_ptr()))

    def state_dict(self):
        """torch.optim.SGD.state_dict() layout for the reference's parameter list (momentum 0: torch keeps an EMPTY
        state, so does this)."""
        state = {}
        if self.step_count > 0 and self.flat_m is not None:
            for i, (p, off) in zip(self.param_ids, self._offsets()):
                state[i] = {"momentum_buffer": self.flat_m[off:off + p.numel()].view_as(p).clone()}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay,
                 "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(self.n_param_ids))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != self.n_param_ids:
            raise ValueError(f"optimizer state has {len(ids)} parameters, this optimiser's id space {self.n_param_ids}")
        g0 = sd["param_groups"][0]
        if float(g0.get("momentum", 0.0)) != self.momentum:
            raise ValueError(f"optimizer state was written with momentum {g0.get('momentum')}, this one has {self.momentum}")
        self.lr, self.weight_decay = g0["lr"], g0.get("weight_decay", 0.0)
        seen = False
        with torch.no_grad():
            for k, (p, off) in zip(self.param_ids, self._offsets()):
                st = sd["state"].get(ids[k])
                if st is None:
                    continue
                seen = True
                if self.flat_m is not None and st.get("momentum_buffer") is not None:
                    self.flat_m[off:off + p.numel()].view_as(p).copy_(st["momentum_buffer"])
        # torch's SGD keeps no step count; the fused kernel only needs "first step or not" for the velocity (momentum 0:
        # the state is empty and the count is irrelevant to the update)
        self.step_count = 1 if seen else 0
