"""world_size-2 gloo tests (CPU) of the data-parallel plumbing used by bench.py --gpus N:
rank-strided protein sharding, ONE flat gradient all-reduce(avg), BN-buffer broadcast from rank 0, fused
[3, N_L] TP/FN/FP all-reduce.  The same code runs over RCCL on the GPU node (backend "nccl")."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from protnote_amd.utils import distributed as D

    r, _, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    class Opt:  # stand-in for FusedClipAdam: only the flat gradient buffer matters here
        flat_g = torch.arange(10, dtype=torch.float32) * (rank + 1)

    D.allreduce_gradients(Opt)
    model = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        model[1].running_mean.fill_(float(rank + 1))
        model[1].running_var.fill_(10.0 * (rank + 1))
    D.broadcast_buffers(model)
    counts = torch.full((3, 7), float(rank + 1))
    D.allreduce_counts(counts)
    shard = D.shard_batch(11, rank, world)
    q.put((rank, Opt.flat_g.clone(), model[1].running_mean.clone(), model[1].running_var.clone(), counts, shard))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_g = torch.arange(10, dtype=torch.float32) * 1.5  # mean of x*1 and x*2
    seen = []
    for rank, g, rm, rv, counts, shard in res:
        assert torch.allclose(g, want_g)
        assert torch.all(rm == 1.0) and torch.all(rv == 10.0)  # rank 0's buffers everywhere
        assert torch.all(counts == 3.0)
        seen += shard
    assert sorted(seen) == list(range(11))  # shards partition the batch
    assert res[0][5] == [0, 2, 4, 6, 8, 10] and res[1][5] == [1, 3, 5, 7, 9]
