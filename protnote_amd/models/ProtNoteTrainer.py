"""The per-batch arithmetic of the reference trainer on MI355X (protnote/models/ProtNoteTrainer.py):
calculate_tp_fn_fp (:61-83), calculate_f1 (:54-58), calculate_f1_micro (:42-51) and the body of the train
step (:728-770).  The epoch loop, checkpoint cadence, W&B and samplers are outside the hot path."""
import torch

from .. import _lib as L


def calculate_tp_fn_fp(probs, labels, threshold=0.5):
    """Per-label true positives / false negatives / false positives (f32 holding exact integer counts)."""
    L.require_hip(probs, labels)
    B, N = probs.shape
    p = probs.detach().float().contiguous()
    tf = ti = None
    if labels.dtype == torch.int64:
        ti = labels.contiguous()
    else:
        tf = labels.detach().float().contiguous()
    tp = torch.empty(N, dtype=torch.float32, device=p.device)
    fn = torch.empty_like(tp)
    fp = torch.empty_like(tp)
    L.check(L.lib().pn_tp_fn_fp(L.ptr(p), L.ptr(tf), L.ptr(ti), B, N, float(threshold), L.ptr(tp), L.ptr(fn),
                                L.ptr(fp), L.stream_ptr()))
    return tp, fn, fp


def calculate_f1(tp, fn, fp):
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    return 2 * (precision * recall) / (precision + recall + 1e-8)


def calculate_f1_micro(total_tp_per_label, total_fn_per_label, total_fp_per_label):
    return calculate_f1(total_tp_per_label.sum(), total_fn_per_label.sum(), total_fp_per_label.sum())


def train_step(model, loss_fn, optimizer, batch, *, world_size=1, counts=None, threshold=0.5,
               gradient_accumulation_steps=1, batch_idx=0):
    """One optimisation step: forward, loss, backward, (data-parallel gradient all-reduce), clip + Adam,
    per-label TP/FN/FP accumulation into `counts` ([3, N] f32) when given.  Returns the detached loss.
    GRADIENT_ACCUMULATION_STEPS (ProtNoteTrainer.py:732-755): the loss is divided by the number of accumulation steps,
    gradients add up in the flat buffer, and all-reduce + clip + Adam + zero_grad run on every
    `gradient_accumulation_steps`-th batch (`batch_idx` counts from 0)."""
    from ..utils.distributed import allreduce_gradients, broadcast_buffers

    if world_size > 1:
        broadcast_buffers(model)
    if counts is not None and hasattr(loss_fn, "metric_counts"):
        loss_fn.metric_counts, loss_fn.decision_threshold = counts, threshold  # counted inside the loss pass
    logits, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                      label_embeddings=batch["label_embeddings"],
                      label_token_counts=batch.get("label_token_counts"))
    loss = loss_fn(logits, batch["label_multihots"])
    if gradient_accumulation_steps > 1:
        loss = loss / gradient_accumulation_steps
    loss.backward()
    if (batch_idx + 1) % gradient_accumulation_steps == 0:
        if world_size > 1:
            allreduce_gradients(optimizer)
        optimizer.step()
        optimizer.zero_grad()
    if counts is not None and not hasattr(loss_fn, "metric_counts"):  # foreign loss module: separate pass
        with torch.no_grad():
            tp, fn, fp = calculate_tp_fn_fp(torch.sigmoid(logits.detach()), batch["label_multihots"], threshold)
            counts[0] += tp
            counts[1] += fn
            counts[2] += fp
    return loss.detach()


class Trainer:
    """Minimal epoch driver around the fused step (reference ProtNoteTrainer.train_one_epoch :675-825 and
    evaluate :449-673, without W&B / checkpoint cadence / threshold search): per-batch train_step with the loss pass
    counting TP/FN/FP, ONE fused [3, N_L] all-reduce per epoch (the reference does three dist.reduce calls,
    :637-639/:795-797), F1 macro/micro from the counts, and - for evaluation - mAP from the collected logits."""

    def __init__(self, model, loss_fn, optimizer=None, world_size=1, threshold=0.5, gradient_accumulation_steps=1):
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.world_size, self.threshold = world_size, threshold
        self.gradient_accumulation_steps = gradient_accumulation_steps

    def _metrics(self, counts, loss_sum, n_batches):
        from ..utils.distributed import allreduce_counts

        counts = allreduce_counts(counts)
        tp, fn, fp = counts[0], counts[1], counts[2]
        return {"loss": loss_sum / max(n_batches, 1), "f1_macro": float(calculate_f1(tp, fn, fp).mean()),
                "f1_micro": float(calculate_f1_micro(tp, fn, fp))}

    def train_one_epoch(self, loader):
        self.model.train()
        counts, loss_sum, n = None, 0.0, 0
        for batch in loader:
            if counts is None:
                counts = torch.zeros(3, batch["label_multihots"].shape[1], dtype=torch.float32,
                                     device=batch["label_multihots"].device)
            loss_sum += float(train_step(self.model, self.loss_fn, self.optimizer, batch, world_size=self.world_size,
                                         counts=counts, threshold=self.threshold,
                                         gradient_accumulation_steps=self.gradient_accumulation_steps, batch_idx=n))
            n += 1
        return self._metrics(counts, loss_sum, n)

    @torch.no_grad()
    def evaluate(self, loader, with_map=True, estimate_map=False, map_thresholds=50):
        """`estimate_map` follows ESTIMATE_MAP (ProtNoteTrainer.py:477-489): False -> exact AUPRC, True -> binned
        AUPRC with `map_thresholds` thresholds; `with_map=False` is ESTIMATE_MAP: None.  Either way the scores never
        leave the device (the reference moves every batch to the CPU, :540-543)."""
        from ..utils.evaluation import DeviceAveragePrecision, DeviceBinnedAUPRC

        self.model.eval()
        counts, loss_sum, n, ap = None, 0.0, 0, None
        for batch in loader:
            y = batch["label_multihots"]
            if counts is None:
                counts = torch.zeros(3, y.shape[1], dtype=torch.float32, device=y.device)
            logits, _ = self.model(sequence_onehots=batch["sequence_onehots"],
                                   sequence_lengths=batch["sequence_lengths"],
                                   label_embeddings=batch["label_embeddings"])
            if hasattr(self.loss_fn, "metric_counts"):
                self.loss_fn.metric_counts, self.loss_fn.decision_threshold = counts, self.threshold
            loss_sum += float(self.loss_fn(logits, y))
            n += 1
            if with_map:
                if ap is None:
                    if estimate_map:
                        ap = DeviceBinnedAUPRC(y.shape[1], y.device, threshold=map_thresholds)
                    else:
                        total = len(loader.dataset) if hasattr(loader, "dataset") else 0
                        ap = DeviceAveragePrecision(y.shape[1], max(total, y.shape[0]), y.device, growable=True)
                ap.update(torch.sigmoid(logits), y)   # the reference scores probabilities (:521-523)
        out = self._metrics(counts, loss_sum, n)
        if ap is not None:
            m = ap.compute()
            out.update(map_micro=m["map_micro"], map_macro=m["map_macro"])
        return out
