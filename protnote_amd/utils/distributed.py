"""Data-parallel plumbing for one-process-per-GPU runs over RCCL/xGMI (reference bin/main.py:192-200,452 and
ProtNoteTrainer.py:637-639,795-797).  Proteins are sharded across ranks; labels, label embeddings, weights and
Adam state are replicated; BatchNorm statistics stay per-rank (SYNC_BN False).  Collectives per step:
ONE all-reduce(avg) of the flat gradient buffer and ONE broadcast of the flat BN-buffer block (DDP's
broadcast_buffers); per epoch ONE all-reduce of the fused [3, N_L] TP/FN/FP block."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torch.distributed init from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # one process per GPU; PN_SHARE_GPU=1 lets several ranks share device 0 (single-GPU dry runs over gloo)
        torch.cuda.set_device(0 if os.environ.get("PN_SHARE_GPU") == "1" else local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("PN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)  # "nccl" is RCCL on ROCm
    return rank, local, world


def shard_batch(n_items: int, rank: int, world: int):
    """Rank-strided protein shard, as the reference samplers do (samplers.py:61,111)."""
    return list(range(rank, n_items, world))


def _float_buffers(model):
    return [b for _, b in model.named_buffers() if b.is_floating_point()]


def broadcast_buffers(model, src: int = 0):
    """DDP broadcast_buffers equivalent: rank `src`'s BN running statistics replace everyone's, as ONE
    flat broadcast."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bufs = _float_buffers(model)
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.broadcast(flat, src=src)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


def allreduce_gradients(optimizer):
    """Average the flat gradient buffer of a FusedClipAdam across ranks (one RCCL all-reduce over xGMI)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.all_reduce(optimizer.flat_g, op=dist.ReduceOp.SUM)
    optimizer.flat_g.mul_(1.0 / dist.get_world_size())


def allreduce_counts(counts):
    """counts: [3, N_L] f32 block of per-label TP/FN/FP (reference does three dist.reduce calls)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
