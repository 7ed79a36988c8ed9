"""Loss factory on MI355X - drop-in twin of protnote/utils/losses.py::get_loss (reference :270-294).

BCE (BCEWithLogitsLoss(reduction='mean', pos_weight), reference :275-276) and FocalLoss (reference :171-213,
the shipped default) run as ONE fused HIP pass over the [B, N_L] logits (pn_loss_fwd_bwd) that produces the
mean loss and d loss / d logits together; backward just scales the cached gradient.  The reference's other
BCE variants ride on the same pass: BatchWeightedBCE and WeightedBCE / CBLoss as element weights, RGDBCE as a
rescale by a device-side scalar.  SupCon (unused in the reference) has its own row-softmax kernel (pn_supcon_fwd_bwd)."""
import torch

from .. import _lib as L

_BCE, _FOCAL = 0, 1


class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, kind, pos_weight, gamma, alpha, smoothing, counts=None, threshold=0.5,
                weight_mode=0, label_weights=None, rgd_temperature=-1.0):
        L.require_hip(logits, target)
        if logits.dim() != 2 or target.shape != logits.shape:
            raise ValueError("expected logits and targets of the same [B, N] shape")
        x = logits.detach().float().contiguous()
        B, N = x.shape
        tgt, tkind = L.typed_targets(target)  # int64 (the reference collator's dtype), uint8 / bool (1 B per pair) or float32
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dlog = torch.empty_like(x)
        ws = L.workspace(L.lib().pn_loss_ws_bytes(B, N), x.device, "loss")
        lw = None
        if label_weights is not None:
            lw = label_weights.detach().to(device=x.device, dtype=torch.float32).contiguous()
            if lw.numel() != N:
                raise ValueError(f"label_weights must have one entry per label ({N}), got {lw.numel()}")
        tp = fn = fp = None
        if counts is not None:  # [3, N] f32 accumulators of per-label TP / FN / FP (ProtNoteTrainer.py:61-83)
            if counts.shape != (3, N) or counts.dtype != torch.float32 or not counts.is_contiguous():
                raise ValueError("metric counts must be a contiguous float32 [3, N_labels] tensor")
            tp, fn, fp = counts[0], counts[1], counts[2]
        L.check(L.lib().pn_loss_fwd_bwd_t(L.ptr(x), L.ptr(tgt), tkind, B, N, kind, float(pos_weight), float(gamma),
                                          float(alpha), float(smoothing), float(threshold), L.ptr(loss), L.ptr(dlog),
                                          L.ptr(tp), L.ptr(fn), L.ptr(fp), int(weight_mode), L.ptr(lw),
                                          float(rgd_temperature), L.ptr(ws), ws.numel(), L.stream_ptr()))
        ctx.dlog = dlog
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.dlog * grad_out
        ctx.dlog = None
        return (g,) + (None,) * 11


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


class RGDBCE(_CountsMixin, torch.nn.Module):
    """Reference losses.py:58-75.  As executed there (legacy `reduce="none"` = mean reduction) the re-weighting acts on
    the scalar mean loss: m * exp(min(m, T) / (T + 1)), factor detached."""

    def __init__(self, temperature: float):
        super().__init__()
        assert temperature is not None, "temperature must be provided and not None"
        self.temperature = float(temperature)

    def forward(self, input, target):
        return _FusedLossFn.apply(input, target, _BCE, 1.0, 0.0, -1.0, 0.0, self.metric_counts,
                                  self.decision_threshold, 0, None, self.temperature)


class BatchWeightedBCE(_CountsMixin, torch.nn.Module):
    """Reference losses.py:124-146: positives and negatives of the batch weigh total/2 each (epsilon 1e-10)."""

    def __init__(self, epsilon=1e-10):
        super().__init__()
        if epsilon != 1e-10:
            raise NotImplementedError("BatchWeightedBCE: only the reference's epsilon=1e-10 is built into the kernel")

    def forward(self, input, target):
        return _FusedLossFn.apply(input, target, _BCE, 1.0, 0.0, -1.0, 0.0, self.metric_counts,
                                  self.decision_threshold, 1, None, -1.0)


class WeightedBCE(_CountsMixin, torch.nn.Module):
    """Reference losses.py:109-121: every element of row i weighs sum_j label_weights[j] * target[i, j]."""

    def __init__(self, label_weights: torch.Tensor):
        super().__init__()
        assert label_weights is not None, "label_weights must be provided and not None"
        self.label_weights = label_weights

    def forward(self, input, target):
        return _FusedLossFn.apply(input, target, _BCE, 1.0, 0.0, -1.0, 0.0, self.metric_counts,
                                  self.decision_threshold, 2, self.label_weights, -1.0)


class CBLoss(WeightedBCE):
    """Reference losses.py:78-106: class-balanced label weights (1 - beta) / (1 - beta^n_j) (infinite effective number
    where it would be 0), normalised to sum to the number of classes, then the WeightedBCE row weighting."""

    def __init__(self, label_weights: torch.Tensor, beta=0.9999):
        assert label_weights is not None, "label_weights must be provided and not None"
        counts = torch.as_tensor(label_weights).float()
        eff = 1.0 - torch.pow(torch.tensor(beta), counts)
        eff = torch.where(eff == 0, torch.tensor(float("inf")), eff)
        w = (1.0 - beta) / eff
        super().__init__(w / torch.sum(w) * len(counts))
        self.beta = beta


class _SupConFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        L.require_hip(logits, target)
        if logits.dim() != 2 or target.shape != logits.shape:
            raise ValueError("expected logits and targets of the same [B, N] shape")
        x = logits.detach().float().contiguous()
        B, N = x.shape
        tf = ti = None  # (pn_supcon_fwd_bwd keeps the two-pointer form: int64 or float32 targets)
        if target.dtype == torch.int64:
            ti = target.contiguous()
        else:
            tf = target.detach().float().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dlog = torch.empty_like(x)
        ws = L.workspace(L.lib().pn_supcon_ws_bytes(B), x.device, "loss")
        L.check(L.lib().pn_supcon_fwd_bwd(L.ptr(x), L.ptr(tf), L.ptr(ti), B, N, L.ptr(loss), L.ptr(dlog), L.ptr(ws),
                                          ws.numel(), L.stream_ptr()))
        ctx.dlog = dlog
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.dlog * grad_out
        ctx.dlog = None
        return g, None


class SupCon(torch.nn.Module):
    """Reference losses.py:7-56 (marked "NOT CURRENTLY USING THIS" there): mean over proteins of the mean log-softmax
    (over the label axis) of their positive labels.  Like the reference's forward it never reads `temperature`; a protein
    without positives contributes 0 to the loss and NaN to the gradient (the reference's nan_to_num quirk)."""

    def __init__(self, temperature: float, base_temperature=0.07):
        super().__init__()
        assert temperature is not None, "temperature must be provided and not None"
        self.temperature, self.base_temperature = temperature, base_temperature

    def forward(self, input, target):
        return _SupConFn.apply(input, target)


def get_loss(config: dict, label_weights: torch.Tensor = None, bce_pos_weight: torch.Tensor = None):
    name = config["params"]["LOSS_FN"]
    if name == "BCE":
        return BCEWithLogitsLoss(pos_weight=bce_pos_weight)
    if name == "FocalLoss":
        return FocalLoss(gamma=config["params"]["FOCAL_LOSS_GAMMA"], alpha=config["params"]["FOCAL_LOSS_ALPHA"],
                         label_smoothing=config["params"]["LABEL_SMOOTHING"])
    if name == "WeightedBCE":
        return WeightedBCE(label_weights=label_weights)
    if name == "CBLoss":
        return CBLoss(label_weights=label_weights)
    if name == "BatchWeightedBCE":
        return BatchWeightedBCE()
    if name == "RGDBCE":
        return RGDBCE(temperature=config["params"]["RGDBCE_TEMP"])
    if name == "SupCon":
        return SupCon(temperature=config["params"]["SUPCON_TEMP"])
    raise ValueError(f"Unknown loss function {name}")
