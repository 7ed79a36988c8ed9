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
    tgt, tkind = L.typed_targets(labels)
    tp = torch.empty(N, dtype=torch.float32, device=p.device)
    fn = torch.empty_like(tp)
    fp = torch.empty_like(tp)
    L.check(L.lib().pn_tp_fn_fp_t(L.ptr(p), L.ptr(tgt), tkind, B, N, float(threshold), L.ptr(tp), L.ptr(fn),
                                  L.ptr(fp), L.stream_ptr()))
    return tp, fn, fp


def calculate_f1(tp, fn, fp):
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    return 2 * (precision * recall) / (precision + recall + 1e-8)


def calculate_f1_micro(total_tp_per_label, total_fn_per_label, total_fp_per_label):
    return calculate_f1(total_tp_per_label.sum(), total_fn_per_label.sum(), total_fp_per_label.sum())


def train_step(model, loss_fn, optimizer, batch, *, world_size=1, counts=None, threshold=0.5,
               gradient_accumulation_steps=1, batch_idx=0, is_last_batch=False):
    """One optimisation step: forward, loss, backward, (data-parallel gradient all-reduce), clip + Adam,
    per-label TP/FN/FP accumulation into `counts` ([3, N] f32) when given.  Returns the detached loss.
    GRADIENT_ACCUMULATION_STEPS (ProtNoteTrainer.py:732-755): the loss is divided by the number of accumulation steps,
    gradients add up in the flat buffer, and all-reduce + clip + Adam + zero_grad run on every
    `gradient_accumulation_steps`-th batch (`batch_idx` counts from 0; the reference's counter is
    `training_step = batch_idx + 1`, :687,741) and on the last batch of an epoch (`is_last_batch`, :741-743), so no
    partial window leaks into the next epoch."""
    from ..utils.distributed import allreduce_gradients, broadcast_buffers

    if world_size > 1:
        broadcast_buffers(model)
    if counts is not None and hasattr(loss_fn, "metric_counts"):
        loss_fn.metric_counts, loss_fn.decision_threshold = counts, threshold  # counted inside the loss pass
    # tokenized_labels: LABEL_EMBEDDING_POOLING_METHOD 'all' reads its attention mask (ProtNoteTrainer.py:712-729 passes the
    # whole batch dict to the model)
    logits, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                      label_embeddings=batch["label_embeddings"],
                      label_token_counts=batch.get("label_token_counts"),
                      tokenized_labels=batch.get("tokenized_labels"))
    loss = loss_fn(logits, batch["label_multihots"])
    if gradient_accumulation_steps > 1:
        loss = loss / gradient_accumulation_steps
    loss.backward()
    if (batch_idx + 1) % gradient_accumulation_steps == 0 or is_last_batch:
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
    :637-639/:795-797), F1 macro/micro from the counts, and - for evaluation - mAP from the collected logits.
    The per-batch losses are summed on the device: no host synchronisation inside an epoch."""

    def __init__(self, model, loss_fn, optimizer=None, world_size=1, threshold=0.5, gradient_accumulation_steps=1):
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.world_size, self.threshold = world_size, threshold
        self.gradient_accumulation_steps = gradient_accumulation_steps
        self.training_step = 0  # global batch counter across epochs (reference :687,851)
        if world_size > 1:  # what wrapping in DistributedDataParallel does (bin/main.py:452): start from rank 0's state
            from ..utils.distributed import sync_initial_state

            sync_initial_state(model, optimizer)

    def _metrics(self, counts, loss_sum, n_batches):
        from ..utils.distributed import allreduce_counts, allreduce_mean_loss

        counts = allreduce_counts(counts)
        tp, fn, fp = counts[0], counts[1], counts[2]
        # sync_and_compute(avg_loss) (:655,812): the mean over ALL ranks' batches, not the local one
        loss = allreduce_mean_loss(float(loss_sum), n_batches, counts.device)
        return {"loss": loss, "f1_macro": float(calculate_f1(tp, fn, fp).mean()),
                "f1_micro": float(calculate_f1_micro(tp, fn, fp))}

    def train_one_epoch(self, loader):
        self.model.train()
        counts, loss_sum, n = None, None, 0
        try:
            n_total = len(loader)
        except TypeError:
            n_total = None
        it = iter(loader)
        batch = next(it, None)
        while batch is not None:
            nxt = next(it, None)  # look-ahead: the last batch of the epoch always steps (reference :741-743)
            if counts is None:
                dev = batch["label_multihots"].device
                counts = torch.zeros(3, batch["label_multihots"].shape[1], dtype=torch.float32, device=dev)
                loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
            last = nxt is None or (n_total is not None and n + 1 == n_total)
            loss = train_step(self.model, self.loss_fn, self.optimizer, batch, world_size=self.world_size,
                              counts=counts, threshold=self.threshold,
                              gradient_accumulation_steps=self.gradient_accumulation_steps,
                              batch_idx=self.training_step, is_last_batch=last)
            self.training_step += 1
            loss_sum += loss.double()
            n += 1
            batch = nxt
        return self._metrics(counts, loss_sum, n)

    @torch.no_grad()
    def evaluate(self, loader, with_map=True, estimate_map=False, map_thresholds=50, represented_label_mask=None):
        """`estimate_map` follows ESTIMATE_MAP (ProtNoteTrainer.py:477-489): False -> exact AUPRC, True -> binned
        AUPRC with `map_thresholds` thresholds; `with_map=False` is ESTIMATE_MAP: None.  Either way the scores never
        leave the device (the reference moves every batch to the CPU, :540-543).  `represented_label_mask` ([N_L] bool,
        the dataset's represented_vocabulary_mask) is the reference's `only_represented_labels=True` (:469-472,517-519):
        loss, counts and mAP are taken over those label columns only.  Leaves the model in train mode, as the reference
        does (:671)."""
        from ..utils.evaluation import DeviceAveragePrecision, DeviceBinnedAUPRC

        self.model.eval()
        counts, loss_sum, n, ap = None, None, 0, None
        keep = None
        for batch in loader:
            y = batch["label_multihots"]
            logits, _ = self.model(sequence_onehots=batch["sequence_onehots"],
                                   sequence_lengths=batch["sequence_lengths"],
                                   label_embeddings=batch["label_embeddings"],
                                   tokenized_labels=batch.get("tokenized_labels"))
            if represented_label_mask is not None:
                if keep is None:
                    keep = torch.as_tensor(represented_label_mask, dtype=torch.bool, device=y.device).nonzero().flatten()
                logits, y = logits.index_select(1, keep).contiguous(), y.index_select(1, keep).contiguous()
            if counts is None:
                counts = torch.zeros(3, y.shape[1], dtype=torch.float32, device=y.device)
                loss_sum = torch.zeros((), dtype=torch.float64, device=y.device)
            if hasattr(self.loss_fn, "metric_counts"):
                self.loss_fn.metric_counts, self.loss_fn.decision_threshold = counts, self.threshold
            loss_sum += self.loss_fn(logits, y).detach().double()
            n += 1
            if with_map:
                if ap is None:
                    if estimate_map:
                        ap = DeviceBinnedAUPRC(y.shape[1], y.device, threshold=map_thresholds)
                    else:
                        total = len(loader.dataset) if hasattr(loader, "dataset") else 0
                        if self.world_size > 1:  # a rank only scores its 1/W shard of the set (the accumulator can grow)
                            total = -(-total // self.world_size)
                        ap = DeviceAveragePrecision(y.shape[1], max(total, y.shape[0]), y.device, growable=True)
                ap.update(torch.sigmoid(logits), y)   # the reference scores probabilities (:521-523)
        out = self._metrics(counts, loss_sum, n)
        if ap is not None:
            m = ap.compute()
            out.update(map_micro=m["map_micro"], map_macro=m["map_macro"])
        self.model.train()
        return out
