"""Batch assembly with the layout contract of protnote/data/collators.py::collate_variable_sequence_length
(reference :5-155): right-pad one-hots with zeros to the batch maximum, stack lengths / multihots, take label
embeddings and token counts from batch[0] only, optional label sub-sampling (randperm / arange / in-batch /
grid / per-rank partition).  The produced dict is exactly what ProtNote.forward / the trainer consume:

    sequence_onehots [B, A, Lmax] f32 | sequence_ids list[str] | sequence_lengths [B] i64
    label_embeddings [N, d] f32       | label_token_counts [N] i64 | label_multihots [B, N] i64
"""
from typing import Dict, List

import torch


def sample_label_indices(batch: List[Dict], label_sample_size=None, distribute_labels=False, shuffle_labels=False,
                         in_batch_sampling=False, grid_sampler=False, world_size=1, rank=0):
    """Index selection of reference collators.py:56-98 (None = keep every label)."""
    if grid_sampler:
        assert label_sample_size is not None, "Must provide label_sample_size if using grid sampler"
        assert not in_batch_sampling, "Can't use in batch sampling with grid sampler"
    else:
        assert not (in_batch_sampling and (label_sample_size is not None)), \
            "Cant use both in_batch_sampling with lable_sample_size"
    num_labels = batch[0]["label_multihots"].shape[0]
    if label_sample_size:
        if grid_sampler:
            return batch[0]["label_idxs"]
        if not distribute_labels:
            return torch.randperm(num_labels)[:label_sample_size] if shuffle_labels else torch.arange(label_sample_size)
        per_part = num_labels // world_size
        part = torch.arange(rank * per_part, rank * per_part + per_part)
        return part[torch.randperm(len(part))[: label_sample_size // world_size]]
    if in_batch_sampling:
        return torch.where(sum(row["label_multihots"] for row in batch) > 0)[0]
    return None


def collate_variable_sequence_length(batch: List[Dict], label_sample_size=None, distribute_labels=False,
                                     shuffle_labels=False, in_batch_sampling=False, grid_sampler=False,
                                     return_label_multihots=True, world_size=1, rank=0):
    max_length = int(max(int(row["sequence_length"]) for row in batch))
    idx = sample_label_indices(batch, label_sample_size, distribute_labels, shuffle_labels, in_batch_sampling,
                               grid_sampler, world_size, rank)
    first = batch[0]
    label_embeddings = first["label_embeddings"] if idx is None else first["label_embeddings"][idx]
    dim = first["sequence_onehots"].shape[0]
    onehots = torch.zeros(len(batch), dim, max_length, dtype=torch.float32)
    for b, row in enumerate(batch):
        n = int(row["sequence_length"])
        onehots[b, :, :n] = row["sequence_onehots"]
    out = {
        "sequence_onehots": onehots,
        "sequence_ids": [row["sequence_id"] for row in batch],
        "sequence_lengths": torch.stack([torch.as_tensor(row["sequence_length"]) for row in batch]),
        "label_embeddings": label_embeddings,
        "label_token_counts": first["label_token_counts"],
    }
    if return_label_multihots:
        out["label_multihots"] = torch.stack(
            [row["label_multihots"] if idx is None else row["label_multihots"][idx] for row in batch])
    return out


def collate_to_device(batch: List[Dict], device, label_sample_size=None, distribute_labels=False, shuffle_labels=False,
                      in_batch_sampling=False, grid_sampler=False, return_label_multihots=True, world_size=1, rank=0,
                      multihot_dtype=None, residue_ids=False):
    """Same batch dict as collate_variable_sequence_length, assembled ON THE DEVICE: the host ships the ragged
    residue indices as uint8 (B*L bytes, one copy) plus offsets, and pn_onehot_batch writes the zero-padded f32
    one-hots [B, A, Lmax] and the lengths in HBM.  Examples may carry `sequence_ints` (residue indices) or the
    reference's `sequence_onehots` [A, L] (argmax'ed here)."""
    import numpy as np

    from .. import _lib as L

    idx = sample_label_indices(batch, label_sample_size, distribute_labels, shuffle_labels, in_batch_sampling,
                               grid_sampler, world_size, rank)
    first = batch[0]
    A = int(first["sequence_onehots"].shape[0]) if "sequence_onehots" in first else int(first["alphabet_size"])
    ints = [np.asarray(r["sequence_ints"], dtype=np.uint8) if "sequence_ints" in r
            else r["sequence_onehots"].argmax(0).to(torch.uint8).numpy() for r in batch]
    lens = [len(v) for v in ints]
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    flat = torch.from_numpy(np.concatenate(ints))
    B, Lmax = len(batch), int(max(lens))
    flat_d = flat.to(device, non_blocking=True)
    off_d = offsets.to(device, non_blocking=True)
    if residue_ids:
        # opt-in: hand the encoder the ids themselves (ResidueIds; pn_encoder_fwd_ids) - no [B, A, Lmax] f32 tensor is written.
        # The default below is the reference collator's dict (row (a)14): f32 one-hots
        from ..models.protein_encoders import ResidueIds

        onehots = ResidueIds(flat_d, off_d, A, Lmax)
        lengths = torch.tensor(lens, dtype=torch.int64).to(device, non_blocking=True)
    else:
        onehots = torch.empty(B, A, Lmax, dtype=torch.float32, device=device)
        lengths = torch.empty(B, dtype=torch.int64, device=device)
        L.check(L.lib().pn_onehot_batch(L.ptr(flat_d), L.ptr(off_d), B, A, Lmax, L.ptr(onehots), L.ptr(lengths),
                                        L.stream_ptr()))
    emb = first["label_embeddings"] if idx is None else first["label_embeddings"][idx]
    out = {"sequence_onehots": onehots, "sequence_ids": [r["sequence_id"] for r in batch],
           "sequence_lengths": lengths, "label_embeddings": emb.to(device, non_blocking=True),
           "label_token_counts": first["label_token_counts"].to(device, non_blocking=True)}
    if return_label_multihots:
        mh = torch.stack([r["label_multihots"] if idx is None else r["label_multihots"][idx] for r in batch])
        # multihot_dtype=torch.uint8: 1 B per pair over PCIe and in the loss / metric kernels (pn_loss_fwd_bwd_t, PN_LABEL_U8)
        # instead of the reference collator's int64 (collators.py:136-137), which stays the default (row (a)14's contract)
        if multihot_dtype is not None:
            mh = mh.to(multihot_dtype)
        out["label_multihots"] = mh.to(device, non_blocking=True)
    return out
