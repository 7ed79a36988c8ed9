"""Loss factory on MI355X - drop-in twin of protnote/utils/losses.py::get_loss (reference :270-294).

BCE (BCEWithLogitsLoss(reduction='mean', pos_weight), reference :275-276) and FocalLoss (reference :171-213,
the shipped default) run as ONE fused HIP pass over the [B, N_L] logits (pn_loss_fwd_bwd) that produces the
mean loss and d loss / d logits together; backward just scales the cached gradient.  The reference's other
losses (RGDBCE, CBLoss, WeightedBCE, BatchWeightedBCE, SupCon) are non-default ablations outside the hot path
and raise NotImplementedError."""
import torch

from .. import _lib as L

_BCE, _FOCAL = 0, 1


class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, kind, pos_weight, gamma, alpha, smoothing, counts=None, threshold=0.5):
        L.require_hip(logits, target)
        if logits.dim() != 2 or target.shape != logits.shape:
            raise ValueError("expected logits and targets of the same [B, N] shape")
        x = logits.detach().float().contiguous()
        B, N = x.shape
        tf = ti = None
        if target.dtype == torch.int64:
            ti = target.contiguous()
        else:
            tf = target.detach().float().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dlog = torch.empty_like(x)
        ws = L.workspace(256, x.device, "loss")
        tp = fn = fp = None
        if counts is not None:  # [3, N] f32 accumulators of per-label TP / FN / FP (ProtNoteTrainer.py:61-83)
            if counts.shape != (3, N) or counts.dtype != torch.float32 or not counts.is_contiguous():
                raise ValueError("metric counts must be a contiguous float32 [3, N_labels] tensor")
            tp, fn, fp = counts[0], counts[1], counts[2]
        L.check(L.lib().pn_loss_fwd_bwd(L.ptr(x), L.ptr(tf), L.ptr(ti), B, N, kind, float(pos_weight), float(gamma),
                                        float(alpha), float(smoothing), float(threshold), L.ptr(loss), L.ptr(dlog),
                                        L.ptr(tp), L.ptr(fn), L.ptr(fp), L.ptr(ws), ws.numel(), L.stream_ptr()))
        ctx.dlog = dlog
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.dlog * grad_out
        ctx.dlog = None
        return g, None, None, None, None, None, None, None, None


class _CountsMixin:
    """Optional K13+K14 fusion: set `metric_counts` to a [3, N_labels] f32 tensor and the same pass over the logits
    that computes the loss and its gradient also accumulates per-label TP / FN / FP at `decision_threshold`."""
    metric_counts = None
    decision_threshold = 0.5


class BCEWithLogitsLoss(_CountsMixin, torch.nn.Module):
    def __init__(self, pos_weight=None):
        super().__init__()
        if pos_weight is not None and torch.is_tensor(pos_weight) and pos_weight.numel() != 1:
            raise NotImplementedError("per-label pos_weight vectors are not implemented (reference passes a scalar)")
        self.pos_weight = 1.0 if pos_weight is None else float(pos_weight)

    def forward(self, input, target):
        return _FusedLossFn.apply(input, target, _BCE, self.pos_weight, 0.0, -1.0, 0.0, self.metric_counts,
                                  self.decision_threshold)


class FocalLoss(_CountsMixin, torch.nn.Module):
    def __init__(self, alpha: float, gamma: float, reduction="mean", label_smoothing=0.0):
        super().__init__()
        assert (alpha is not None) & (gamma is not None), \
            "Both gamma and alpha must be provided and neither should be None"
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' is implemented")
        self.alpha, self.gamma, self.reduction, self.label_smoothing = alpha, gamma, reduction, label_smoothing

    def forward(self, input, target):
        return _FusedLossFn.apply(input, target, _FOCAL, 1.0, self.gamma, self.alpha, self.label_smoothing,
                                  self.metric_counts, self.decision_threshold)


def get_loss(config: dict, label_weights: torch.Tensor = None, bce_pos_weight: torch.Tensor = None):
    name = config["params"]["LOSS_FN"]
    if name == "BCE":
        return BCEWithLogitsLoss(pos_weight=bce_pos_weight)
    if name == "FocalLoss":
        return FocalLoss(gamma=config["params"]["FOCAL_LOSS_GAMMA"], alpha=config["params"]["FOCAL_LOSS_ALPHA"],
                         label_smoothing=config["params"]["LABEL_SMOOTHING"])
    if name in ("WeightedBCE", "CBLoss", "BatchWeightedBCE", "RGDBCE", "SupCon"):
        raise NotImplementedError(f"LOSS_FN={name} is a non-default ablation outside the MI355X hot path")
    raise ValueError(f"Unknown loss function {name}")
