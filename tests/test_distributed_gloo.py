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

    # DDP-construction semantics: differently seeded ranks end up with rank 0's parameters, Adam state and buffers
    torch.manual_seed(100 + rank)
    m2 = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.BatchNorm1d(5))
    with torch.no_grad():
        m2[1].running_mean.fill_(float(rank))
        m2[1].num_batches_tracked.fill_(7 * (rank + 1))

    class Opt2:  # the flat blocks of a FusedClipAdam over m2[0]'s weight
        params = [m2[0].weight]
        flat_w = torch.full((15,), float(rank + 1))
        flat_m = torch.full((15,), 10.0 * (rank + 1))
        flat_v = torch.full((15,), 100.0 * (rank + 1))
        step_count = 3 * (rank + 1)

    m2[0].weight.data = Opt2.flat_w.view(5, 3)
    v0 = m2[0].bias._version
    D.sync_initial_state(m2, Opt2)
    mean_loss = D.allreduce_mean_loss(2.0 * (rank + 1), rank + 1, torch.device("cpu"))
    state = (m2[0].weight.detach().clone(), m2[0].bias.detach().clone(), m2[1].running_mean.clone(),
             int(m2[1].num_batches_tracked), Opt2.flat_m.clone(), Opt2.flat_v.clone(), Opt2.step_count, mean_loss,
             m2[0].bias._version > v0 or rank == 0)
    import pickle

    # by value: tensors handed to an mp queue travel as shared-memory handles that die with this process
    q.put(pickle.dumps((rank, Opt.flat_g.clone(), model[1].running_mean.clone(), model[1].running_var.clone(), counts,
                        shard, state)))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import pickle

    res = sorted((pickle.loads(q.get(timeout=120)) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_g = torch.arange(10, dtype=torch.float32) * 1.5  # mean of x*1 and x*2
    seen = []
    for rank, g, rm, rv, counts, shard, _ in res:
        assert torch.allclose(g, want_g)
        assert torch.all(rm == 1.0) and torch.all(rv == 10.0)  # rank 0's buffers everywhere
        assert torch.all(counts == 3.0)
        seen += shard
    assert sorted(seen) == list(range(11))  # shards partition the batch
    assert res[0][5] == [0, 2, 4, 6, 8, 10] and res[1][5] == [1, 3, 5, 7, 9]
    # sync_initial_state: everything equals rank 0's; the mean loss is over all ranks' batches ((2 + 4) / (1 + 2))
    s0, s1 = res[0][6], res[1][6]
    for a, b in zip(s0[:3], s1[:3]):
        assert torch.equal(a, b)
    assert torch.all(s1[0] == 1.0) and torch.all(s1[2] == 0.0)
    assert s0[3] == s1[3] == 7 and torch.all(s1[4] == 10.0) and torch.all(s1[5] == 100.0) and s0[6] == s1[6] == 3
    assert abs(s0[7] - 2.0) < 1e-12 and abs(s1[7] - 2.0) < 1e-12
    assert s1[8], "in-place sync must bump the parameter version (packed-weight caches key on it)"


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment re-executes itself under torch.distributed.run with N
    ranks on 127.0.0.1 (it used to run ONE rank and report n_gpus: 1)."""
    import sys

    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _exchange_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from protnote_amd.utils import distributed as D
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    D.init_from_env(backend="gloo")
    # rank r sends r + v + 1 elements to rank v (value = 100 * sender + 10 * receiver + position), one chunk may be empty
    chunks = [torch.arange(rank + v + 1, dtype=torch.int32) + 100 * rank + 10 * v for v in range(world)]
    if rank == 1:
        chunks[0] = chunks[0][:0]
    got = DeviceAveragePrecision._exchange(chunks)
    import pickle

    q.put(pickle.dumps((rank, [g.clone() for g in got])))
    torch.distributed.destroy_process_group()


def test_two_rank_variable_exchange():
    """The all-to-all of the rank-sharded exact AP (label blocks to their owner, key ranges to theirs): uneven and empty
    chunks arrive at the right rank in sender order."""
    import pickle

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((pickle.loads(q.get(timeout=120)) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res[0][1], res[1][1]
    assert r0[0].tolist() == [0] and r0[1].tolist() == []                      # from rank 0: 1 element; from rank 1: empty
    assert r1[0].tolist() == [10, 11] and r1[1].tolist() == [110, 111, 112]    # from rank 0: 2; from rank 1: 3


def test_zero_shot_dealing_balances_eight_ranks():
    """bench.py's configs[4] workload at 8 ranks (no GPU needed for the dealing itself): SEQUENCES are dealt rank-strided
    within every length bucket, weak scaling - every rank holds 512 sequences +-1 per bucket, at least 4 batches, the same
    mix of lengths (padded residues within 6 % of each other), and the ranks partition the set.  Round 2 dealt whole
    batches and left ranks 6 and 7 without work."""
    import bench

    world, per_rank = 8, 512
    shares, seen = [], []
    for r in range(world):
        zb, residues, n_seq, mine = bench.zero_shot_batches(per_rank, 128, r, world, "cpu")
        assert n_seq == world * per_rank
        padded = sum(x.shape[0] * x.shape[2] for x, _ in zb)
        shares.append((mine, len(zb), padded))
        seen.append(sum(int(l.numel()) for _, l in zb))
        for x, l in zb:  # a batch never mixes buckets: its padded length is its bucket
            assert x.shape[2] in bench.BUCKETS and int(l.max()) <= x.shape[2] and x.shape[0] <= 128
    assert sum(seen) == world * per_rank
    assert all(abs(m - per_rank) <= len(bench.BUCKETS) for m, _, _ in shares), shares
    assert min(b for _, b, _ in shares) >= 4, shares
    pads = [p for _, _, p in shares]
    assert max(pads) <= 1.06 * min(pads), pads


def _sgd_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from protnote_amd.utils import distributed as D

    D.init_from_env(backend="gloo")
    out = []
    for momentum in (0.0, 0.9):
        lin = torch.nn.Linear(3, 5)

        class SGDShaped:  # the flat blocks FusedClipSGD keeps: no second moment, no velocity when momentum is 0
            params = [lin.weight]
            flat_w = torch.full((15,), float(rank + 1))
            flat_m = None if momentum == 0.0 else torch.full((15,), 10.0 * (rank + 1))
            flat_v = None
            step_count = 5 * (rank + 1)

            @classmethod
            def state_buffers(cls):
                return [t for t in (cls.flat_w, cls.flat_m, cls.flat_v) if t is not None]

        lin.weight.data = SGDShaped.flat_w.view(5, 3)
        D.sync_initial_state(lin, SGDShaped)  # round 4 crashed here: dist.broadcast(None)
        out.append((SGDShaped.flat_w.clone(), None if SGDShaped.flat_m is None else SGDShaped.flat_m.clone(),
                    SGDShaped.step_count, lin.bias.detach().clone()))
    import pickle

    q.put(pickle.dumps((rank, out)))
    torch.distributed.destroy_process_group()


def test_two_rank_sync_with_sgd_shaped_optimizer():
    """OPTIMIZER: SGD under data parallelism (reference ProtNoteTrainer.py:238-243 + DDP): the optimiser has no second
    moment and, with momentum 0, no velocity block either; sync_initial_state must broadcast only what exists."""
    import pickle

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sgd_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(pickle.loads(q.get(timeout=120)) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for case in range(2):
        w0, m0, s0, b0 = res[0][case]
        w1, m1, s1, b1 = res[1][case]
        assert torch.all(w1 == 1.0) and torch.equal(w0, w1) and s0 == s1 == 5 and torch.equal(b0, b1)
        if case == 0:
            assert m0 is None and m1 is None
        else:
            assert torch.all(m1 == 10.0) and torch.equal(m0, m1)
