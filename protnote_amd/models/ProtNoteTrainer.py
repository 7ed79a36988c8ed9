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


def train_step(model, loss_fn, optimizer, batch, *, world_size=1, counts=None, threshold=0.5):
    """One optimisation step: forward, loss, backward, (data-parallel gradient all-reduce), clip + Adam,
    per-label TP/FN/FP accumulation into `counts` ([3, N] f32) when given.  Returns the detached loss."""
    from ..utils.distributed import allreduce_gradients, broadcast_buffers

    if world_size > 1:
        broadcast_buffers(model)
    if counts is not None and hasattr(loss_fn, "metric_counts"):
        loss_fn.metric_counts, loss_fn.decision_threshold = counts, threshold  # counted inside the loss pass
    logits, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                      label_embeddings=batch["label_embeddings"],
                      label_token_counts=batch.get("label_token_counts"))
    loss = loss_fn(logits, batch["label_multihots"])
    loss.backward()
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
