"""Data-parallel plumbing for one-process-per-GPU runs over RCCL/xGMI (reference bin/main.py:192-200,452 and
ProtNoteTrainer.py:637-639,795-797).  Proteins are sharded across ranks; labels, label embeddings, weights and
Adam state are replicated; BatchNorm statistics stay per-rank (SYNC_BN False).  Collectives:
  * once, when the optimiser is attached: ONE broadcast of rank 0's flat parameter / Adam-moment / BN-buffer blocks
    (what DistributedDataParallel does at construction, bin/main.py:452);
  * per step: ONE all-reduce(avg) of the flat gradient buffer and ONE broadcast of the flat BN-buffer block (DDP's
    broadcast_buffers);
  * per epoch: ONE all-reduce of the fused [3, N_L] TP/FN/FP block and one of (loss sum, batch count).
Every collective is timed with events on the stream it is issued on (`comm_stats()`), so a scaling run shows where
the non-scaling part comes from."""
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


def active() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_batch(n_items: int, rank: int, world: int):
    """Rank-strided protein shard, as the reference samplers do (samplers.py:61,111)."""
    return list(range(rank, n_items, world))


# ---------------------------------------------------------------------------------------------------------------
# per-collective timing (device time between two events on the issuing stream; read by bench.py)
# ---------------------------------------------------------------------------------------------------------------
_stats = {}     # name -> [calls, bytes, [pending (e0, e1)], ms_done]
_timing = False


def comm_timing(on: bool) -> None:
    """Switch event timing of the collectives on/off and clear what was recorded."""
    global _timing
    _timing = bool(on)
    _stats.clear()


class _Timed:
    def __init__(self, name, tensor):
        self.on = _timing and tensor.is_cuda
        self.rec = _stats.setdefault(name, [0, 0, [], 0.0])
        self.rec[0] += 1
        self.rec[1] += tensor.numel() * tensor.element_size()

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            self.rec[2].append((self.e0, self.e1))
        return False


def comm_stats():
    """{name: {"calls", "bytes", "ms"}} - synchronises the device to read the events."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    out = {}
    for name, rec in _stats.items():
        for e0, e1 in rec[2]:
            rec[3] += e0.elapsed_time(e1)
        rec[2] = []
        out[name] = {"calls": rec[0], "bytes": rec[1], "ms": rec[3]}
    return out


# ---------------------------------------------------------------------------------------------------------------
# collectives
# ---------------------------------------------------------------------------------------------------------------
def _float_buffers(model):
    return [b for _, b in model.named_buffers() if b.is_floating_point()]


def _int_buffers(model):
    return [b for _, b in model.named_buffers() if not b.is_floating_point()]


def _broadcast_list(bufs, src, name):
    if not bufs:
        return
    flat = torch.cat([b.detach().reshape(-1) for b in bufs])
    with _Timed(name, flat):
        dist.broadcast(flat, src=src)
    if dist.get_rank() != src:
        with torch.no_grad():
            torch._foreach_copy_(bufs, [c.view_as(b) for b, c in zip(bufs, flat.split([b.numel() for b in bufs]))])


def broadcast_buffers(model, src: int = 0):
    """DDP broadcast_buffers equivalent: rank `src`'s BN running statistics replace everyone's, as ONE flat
    broadcast (one concat, one collective, one multi-tensor copy back on the receiving ranks)."""
    if not active():
        return
    _broadcast_list(_float_buffers(model), src, "bn_buffer_broadcast")


def sync_initial_state(model, optimizer=None, src: int = 0):
    """What DistributedDataParallel does when it wraps the model (reference bin/main.py:452): every rank starts from
    rank `src`'s parameters and buffers.  With a FusedClipAdam the flat weight / moment blocks and the step count
    travel too, so resumed runs agree as well.  Without this, ranks that were seeded differently stay different
    forever (gradients are averaged, parameters are not)."""
    if not active():
        return
    from .optim import _bump_generation

    _bump_generation()  # weights change under caches of weight-derived tensors (the eval-mode L_e cache)
    if optimizer is not None and hasattr(optimizer, "flat_w"):
        bufs = optimizer.state_buffers() if hasattr(optimizer, "state_buffers") else [
            t for t in (getattr(optimizer, n, None) for n in ("flat_w", "flat_m", "flat_v")) if t is not None]
        for t in bufs:  # SGD keeps no second moment, and no velocity when momentum is 0
            with _Timed("initial_state_broadcast", t):
                dist.broadcast(t, src=src)
        step = torch.tensor([optimizer.step_count], dtype=torch.int64, device=optimizer.flat_w.device)
        dist.broadcast(step, src=src)
        optimizer.step_count = int(step.item())
        owned = {p.data_ptr() for p in optimizer.params}
    else:
        owned = set()
    # frozen encoder etc.; the Parameters themselves (not .data) so their version counters move and caches keyed on
    # them (the encoder's packed conv weights) refresh
    rest = [p for p in model.parameters() if p.data_ptr() not in owned]
    for dt in {t.dtype for t in rest}:
        _broadcast_list([t for t in rest if t.dtype == dt], src, "initial_state_broadcast")
    _broadcast_list(_float_buffers(model), src, "initial_state_broadcast")
    _broadcast_list(_int_buffers(model), src, "initial_state_broadcast")


def allreduce_gradients(optimizer):
    """Average the flat gradient buffer of a FusedClipAdam across ranks: ONE RCCL all-reduce over xGMI (the
    averaging is the collective's own AVG reduction on RCCL; gloo has no AVG, so the 1/world scale is a second
    pass there)."""
    if not active():
        return
    g = optimizer.flat_g
    if dist.get_backend() == "nccl":
        with _Timed("grad_allreduce", g):
            dist.all_reduce(g, op=dist.ReduceOp.AVG)
    else:
        with _Timed("grad_allreduce", g):
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            g.mul_(1.0 / dist.get_world_size())


def allreduce_counts(counts):
    """counts: [3, N_L] f32 block of per-label TP/FN/FP (reference does three dist.reduce calls)."""
    if active():
        with _Timed("counts_allreduce", counts):
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def allreduce_mean_loss(loss_sum: float, n_batches: int, device):
    """sync_and_compute(avg_loss) of the reference (ProtNoteTrainer.py:655,812): the mean over every rank's batches."""
    if not active():
        return loss_sum / max(n_batches, 1)
    t = torch.tensor([loss_sum, float(n_batches)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0] / t[1].clamp(min=1.0))


# ---------------------------------------------------------------------------------------------------------------
# SYNC_BN: True (reference bin/main.py:449-450)
# ---------------------------------------------------------------------------------------------------------------
_sync_bn = {}  # keeps the staging tensor and the ctypes callback alive while the library holds their addresses


def enable_sync_batchnorm(device=None):
    """The reference converts every BatchNorm to SyncBatchNorm when SYNC_BN is set (bin/main.py:449-450).  Here the
    BatchNorm statistics are reduced inside the HIP library's calls, so the switch is library-wide: the library hands
    its per-rank column sums to this callback, which all-reduces them over the process group (RCCL), in the forward
    and in the backward of every train-mode BatchNorm (encoder, W_p, W_l, output MLP).  No-op on one rank."""
    import ctypes as C

    from .. import _lib as L

    if not active():
        return False
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    stage = torch.zeros(16384, dtype=torch.float64, device=dev)
    err = []

    def hook(n, _user):
        try:
            with _Timed("sync_bn_allreduce", stage[:n]):
                dist.all_reduce(stage[:n], op=dist.ReduceOp.SUM)
            return 0
        except Exception as e:  # surfaces as a library error ("the all-reduce callback failed") on this rank
            err.append(e)
            return 1

    cb = C.CFUNCTYPE(C.c_int, C.c_long, C.c_void_p)(hook)
    L.check(L.lib().pn_set_sync_bn(C.cast(cb, C.c_void_p), None, L.ptr(stage), stage.numel(), dist.get_world_size()))
    _sync_bn.update(stage=stage, cb=cb, err=err)
    return True


def disable_sync_batchnorm():
    from .. import _lib as L

    L.check(L.lib().pn_set_sync_bn(None, None, None, 0, 1))
    _sync_bn.clear()
