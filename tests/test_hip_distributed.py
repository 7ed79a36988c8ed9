"""Two ranks on ONE MI355X (PN_SHARE_GPU=1, gloo moving the device tensors): the data-parallel code paths with real
HIP buffers - flat-gradient all-reduce inside train_step, BN-buffer broadcast, and the cross-rank evaluation metrics
(exact AP: label-sharded all-to-all for the per-label ranking + a sample sort for the micro ranking, nothing gathered
on every rank; binned AUPRC all-reduces its histograms).  On the 8-GPU node the same
code runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(3)
    n, nl = 601, 37  # odd: the two ranks hold shards of different size
    logits = torch.randn(n, nl, generator=g) * 2
    y = (torch.rand(n, nl, generator=g) < 0.1) | ((logits > 2) & (torch.rand(n, nl, generator=g) < 0.5))
    p = torch.sigmoid(logits).float()
    p[:, :10] = torch.round(p[:, :10] * 20) / 20  # heavy ties: tie groups span both ranks' shards
    p[:, 10] = 0.25                               # one label all tied
    y[:, 5] = False                               # one label without positives (AP = NaN, macro counts it as 0)
    return p, y


def _worker(rank, world, port, q, backend="gloo"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PN_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "gloo":
        os.environ["PN_SHARE_GPU"] = "1"  # both ranks on device 0, gloo moves the device buffers
    else:
        os.environ.pop("PN_SHARE_GPU", None)  # one GPU per rank, RCCL over xGMI
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protnote_amd.models.ProtNoteTrainer import train_step
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils import distributed as D
    from protnote_amd.utils.evaluation import DeviceAveragePrecision, DeviceBinnedAUPRC
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam
    from tests.helpers import make_protnote

    r, _, w = D.init_from_env()
    assert torch.distributed.get_backend() == backend
    dev = f"cuda:{torch.cuda.current_device()}"
    # ---- metrics: each rank holds a strided shard of the proteins
    p, y = _data()
    idx = D.shard_batch(p.shape[0], r, w)
    ap = DeviceAveragePrecision(p.shape[1], len(idx), dev)
    bn = DeviceBinnedAUPRC(p.shape[1], dev, threshold=50)
    ap.update(p[idx].to(dev), y[idx].to(dev))
    bn.update(p[idx].to(dev), y[idx].to(dev))
    m_ap, m_bn = ap.compute(), bn.compute()
    # ---- one data-parallel train step: different batch halves per rank, averaged gradients, rank-0 BN buffers
    g = np.load(os.path.join(GOLDEN, "protnote_small_concatenation.npz"))
    model, _ = make_protnote(g, dev)
    for n_, q_ in model.named_parameters():
        if n_.startswith("sequence_encoder"):
            q_.requires_grad = False
    model.train()
    opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
    x, lens, yy = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]), torch.from_numpy(g["multihots"]).float()
    rows = D.shard_batch(x.shape[0], r, w)
    batch = {"sequence_onehots": x[rows].to(dev), "sequence_lengths": lens[rows].to(dev),
             "label_embeddings": torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(dev),
             "label_multihots": yy[rows].to(dev)}
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    with torch.no_grad():  # make the ranks' weights, Adam moments and BN buffers differ: DDP-construction semantics
        opt.flat_w.mul_(1.0 + 0.05 * r)  # (sync_initial_state) must bring everyone to rank 0's before the step
        opt.flat_m.add_(0.5 * r)
        for b in model.buffers():
            if b.is_floating_point():
                b.add_(0.01 * r)
    w_rank0 = opt.flat_w.clone() if r == 0 else None
    D.sync_initial_state(model, opt)
    if r == 0:
        assert torch.equal(opt.flat_w, w_rank0)
    D.comm_timing(True)
    train_step(model, loss_fn, opt, batch, world_size=w)
    comm = D.comm_stats()
    assert comm["grad_allreduce"]["calls"] == 1 and comm["bn_buffer_broadcast"]["calls"] == 1
    assert comm["grad_allreduce"]["bytes"] == opt.flat_g.numel() * 4
    q.put((r, m_ap["map_micro"], m_ap["map_macro"], m_ap["ap_per_label"].cpu().numpy(), m_bn["map_micro"],
           m_bn["map_macro"], opt.flat_w.cpu().numpy(), float(opt.last_grad_norm.item())))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run_two_ranks(backend):
    from oracle import metrics_oracle as MO

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    p, y = _data()
    per = np.array([MO.average_precision_fast(p[:, j].numpy(), y[:, j].numpy()) for j in range(p.shape[1])])
    micro = MO.average_precision_fast(p.numpy().ravel(), y.numpy().ravel())
    thr = torch.linspace(0, 1.0, 50).numpy()
    bmicro = MO.binned_auprc(p.numpy().ravel(), y.numpy().ravel(), thr)
    bmacro = MO.macro_mean([MO.binned_auprc(p[:, j].numpy(), y[:, j].numpy(), thr) for j in range(p.shape[1])])
    for _, mi, ma, apl, bmi, bma, _w, _gn in res:  # every rank reports the metric of the WHOLE evaluation set
        np.testing.assert_allclose(mi, micro, rtol=1e-12)
        np.testing.assert_allclose(ma, MO.macro_mean(per), rtol=1e-12)
        np.testing.assert_allclose(apl, per, rtol=1e-12)  # NaN (label 5) in the same place
        np.testing.assert_allclose(bmi, bmicro, rtol=1e-12)
        np.testing.assert_allclose(bma, bmacro, rtol=1e-12)
    # averaged gradients + identical start => identical parameters on both ranks after the step
    np.testing.assert_array_equal(res[0][6], res[1][6])
    assert res[0][7] == res[1][7] and np.isfinite(res[0][7])  # same averaged gradient -> same clip norm
    return res


def test_two_ranks_share_one_gpu():
    """1-GPU fallback of the data-parallel test: both ranks on device 0, gloo as the transport."""
    _run_two_ranks("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (RCCL over xGMI)")
def test_two_ranks_rccl():
    """The same data-parallel step and cross-rank metrics with one GPU per rank over RCCL (backend "nccl"), as
    bin/main.py:192-200,452 runs the reference.  The result must agree with the gloo transport: the all-reduce
    averages the same two gradient blocks either way."""
    res = _run_two_ranks("nccl")
    ref = _run_two_ranks("gloo")
    np.testing.assert_allclose(res[0][6], ref[0][6], rtol=0, atol=1e-6)


def _sgd_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PN_SHARE_GPU="1")
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protnote_amd.models.ProtNoteTrainer import train_step
    from protnote_amd.utils import distributed as D
    from protnote_amd.utils.configs import build_training
    from tests.helpers import make_protnote

    r, _, w = D.init_from_env()
    dev = f"cuda:{torch.cuda.current_device()}"
    g = np.load(os.path.join(GOLDEN, "protnote_small_concatenation.npz"))
    x, lens, yy = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]), torch.from_numpy(g["multihots"]).float()
    rows = D.shard_batch(x.shape[0], r, w)
    batch = {"sequence_onehots": x[rows].to(dev), "sequence_lengths": lens[rows].to(dev),
             "label_embeddings": torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(dev),
             "label_multihots": yy[rows].to(dev)}
    out = []
    for momentum in (0.0, 0.9):
        torch.manual_seed(10 + r)  # differently initialised replicas
        model, _ = make_protnote(g, dev)
        with torch.no_grad():
            for q_ in model.parameters():
                q_.mul_(1.0 + 0.05 * r)
        model.train()
        cfg = {"params": {"LOSS_FN": "BCE", "BCE_POS_WEIGHT": 1, "OPTIMIZER": "SGD", "LEARNING_RATE": 1e-2,
                          "WEIGHT_DECAY": 1e-3, "CLIP_VALUE": 1, "TRAIN_SEQUENCE_ENCODER": False}}
        loss_fn, opt, trainer = build_training(cfg, model, world_size=w)  # Trainer.__init__ -> sync_initial_state
        if momentum:
            from protnote_amd.utils.optim import FusedClipSGD

            opt = FusedClipSGD(opt.params, lr=1e-2, momentum=momentum, weight_decay=1e-3, max_norm=1.0)
            with torch.no_grad():
                opt.flat_m.add_(0.25 * r)
            D.sync_initial_state(model, opt)
        assert opt.flat_v is None and (opt.flat_m is None) == (momentum == 0.0)
        for _ in range(2):
            train_step(model, loss_fn, opt, batch, world_size=w)
        sd = opt.state_dict()
        assert (len(sd["state"]) == 0) == (momentum == 0.0)
        out.append(opt.flat_w.cpu().numpy())
    q.put((r, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_sgd_through_build_training():
    """OPTIMIZER: SGD with world_size 2 (reference ProtNoteTrainer.py:238-243 under DDP): Trainer.__init__ syncs the
    initial state of an optimiser that keeps no moment blocks (it crashed on them in round 4), and after two averaged
    steps both replicas hold identical weights - with and without momentum."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sgd_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for case in range(2):
        np.testing.assert_array_equal(res[0][case], res[1][case])
        assert np.isfinite(res[0][case]).all()


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment must run two ranks and say so (n_gpus: 2, dp2, comm
    block).  With >= 2 GPUs this is the real RCCL path; on a 1-GPU box both ranks share device 0 over gloo (dry run of
    the same code: launch, rendezvous, initial-state broadcast, per-step collectives, max-over-ranks timing)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    two = torch.cuda.device_count() >= 2
    if not two:
        env.update(PN_SHARE_GPU="1", PN_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "16", "--seq-len", "64", "--labels", "512", "--no-cpu-baseline",
                          "--zero-shot-seqs", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    # rank 0 prints the full record first ({"bench_detail": ...}) and the compact headline (what the driver parses) LAST
    assert len(lines) == 2 and len(lines[-1]) < 4096, out.stdout[-2000:]
    head = json.loads(lines[-1])
    j = json.loads(lines[0])["bench_detail"]
    assert head["n_gpus"] == 2 and head["config"]["parallelism"] == "dp2" and head["value"] == pytest.approx(j["value"], rel=1e-5)
    assert head["comm"]["rccl_ranks"] == 2 and head["comm"]["replicas_in_sync"] is True
    assert "roofline" in head and "kernels" not in head
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 32
    assert j["comm"]["rccl_ranks"] == 2 and j["comm"]["backend"] == ("nccl" if two else "gloo")
    c = j["comm"]["collectives_rank0"]
    assert c["grad_allreduce"]["calls_per_step"] == 1 and c["bn_buffer_broadcast"]["calls_per_step"] == 1
    assert j["value"] > 0 and j["forward_only"]["f32"]["value"] > 0
    assert all(v["value"] > 0 for v in j["zero_shot"]["f32"].values())
    # after K all-reduced steps from deliberately different seeds the replicas hold bit-identical weights
    assert j["comm"]["replicas_in_sync"] is True and len(j["comm"]["flat_w_checksum_per_rank"]) == 2
    assert len(j["comm"]["own_seconds_before_barrier"]["per_rank"]) == 2
    # configs[4]: sequences (not batches) are dealt to the ranks - nobody idles, the shares differ by at most one
    # sequence per length bucket, and every rank reports its own seconds
    zs = j["zero_shot"]
    assert zs["sequences_per_rank"] == [8.0, 8.0] or (sum(zs["sequences_per_rank"]) == 16 and
                                                      max(zs["sequences_per_rank"]) - min(zs["sequences_per_rank"]) <= 5)
    assert min(zs["batches_per_rank"]) >= 1
    for v in zs["f32"].values():
        assert len(v["rank_seconds"]["per_rank"]) == 2 and v["rank_seconds"]["min"] > 0


def _sync_bn_worker(rank, world, port, q, train_encoder, mode="even"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PN_SHARE_GPU="1")
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils import distributed as D
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam
    from tests.helpers import make_protnote

    r, _, w = D.init_from_env()
    dev = f"cuda:{torch.cuda.current_device()}"
    g = np.load(os.path.join(GOLDEN, "protnote_small_concatenation.npz"))
    x, lens, yy = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]), torch.from_numpy(g["multihots"]).float()
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous()
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))

    def shard(rk):
        if mode == "uneven_B":  # 4 + 2 proteins: the ranks' BatchNorm row counts differ (ragged last batch of an epoch)
            return [0, 1, 2, 3] if rk == 0 else [4, 5]
        if mode == "ragged_L":  # lens [50, 50, 44] | [3, 21, 17]: rank 1's collator pads to 21, not 50 (collators.py:40)
            return [0, 3, 5] if rk == 0 else [1, 2, 4]
        return D.shard_batch(x.shape[0], rk, w)

    def run(rows, sync, reduce=False, lmax=None, weight=1.0):
        model, _ = make_protnote(g, dev, label_embedding_noising_alpha=0.0, train_sequence_encoder=train_encoder)
        params = list(head_parameters(model))
        if train_encoder:
            params += list(model.sequence_encoder.trunk_parameters())
        else:
            for n_, q_ in model.named_parameters():
                if n_.startswith("sequence_encoder"):
                    q_.requires_grad = False
        model.train()
        opt = FusedClipAdam(params, lr=3e-4, max_norm=1.0)
        if sync:
            assert D.enable_sync_batchnorm(dev)
        try:
            xr = x[rows] if lmax is None else x[rows][:, :, :lmax].contiguous()
            logits, _ = model(sequence_onehots=xr.to(dev), sequence_lengths=lens[rows].to(dev),
                              label_embeddings=lab.to(dev))
            loss = loss_fn(logits, yy[rows].to(dev))
            (loss * weight).backward()
            if sync or reduce:
                D.allreduce_gradients(opt)
        finally:
            if sync:
                D.disable_sync_batchnorm()
        torch.cuda.synchronize()
        bufs = {k: v.detach().cpu().clone() for k, v in model.named_buffers()}
        return float(loss), opt.flat_g.detach().cpu().clone(), bufs

    mine = shard(r)
    # unequal shards: the mean of per-rank mean losses is not the whole-batch mean; weigh each rank's loss by its share so
    # that the AVERAGED gradient is the whole-batch gradient (the statistics must then be the whole-batch ones as well)
    wgt = len(mine) * w / x.shape[0]
    lmax = int(lens[mine].max()) if mode == "ragged_L" else None
    l_sync, g_sync, b_sync = run(mine, True, lmax=lmax, weight=wgt)
    l_sync *= wgt
    torch.distributed.barrier()
    _, g_per_rank, _ = run(mine, False, reduce=True, lmax=lmax, weight=wgt)  # control: per-rank statistics
    torch.distributed.barrier()
    out = (r, l_sync, g_sync.numpy(), {k: v.numpy() for k, v in b_sync.items()}, None, g_per_rank.numpy())
    if r == 0:  # the same step on ONE rank over the whole batch, per-rank statistics (= statistics of the whole batch)
        l_full, g_full, b_full = run(shard(0) + shard(1), False)
        out = out[:4] + ((l_full, g_full.numpy(), {k: v.numpy() for k, v in b_full.items()}), out[5])
    import pickle

    q.put(pickle.dumps(out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run_sync_bn(train_encoder, mode):
    import pickle

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_bn_worker, args=(r, world, port, q, train_encoder, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((pickle.loads(q.get(timeout=300)) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_sync_bn_ragged_padding_uses_the_summed_row_count():
    """SYNC_BN with a different padded length on every rank (each collator pads to its own batch maximum,
    collators.py:40): rank 0 holds lens [50, 50, 44] padded to 50, rank 1 lens [3, 21, 17] padded to 21.  The BatchNorm row
    count of the encoder layers is then 150 on one rank and 63 on the other; the global statistics are the summed column
    sums over the SUMMED count (torch.nn.SyncBatchNorm gathers the counts), not over local_count * world.  Checked: both
    ranks end with bit-identical BatchNorm buffers, and the first encoder BatchNorm's running statistics equal the
    oracle's convolution outputs of BOTH shards pooled over 213 rows."""
    from oracle import protnote_oracle as O

    (_, _, _, b0, _, _), (_, _, _, b1, _, _) = _run_sync_bn(False, "ragged_L")
    for k in b0:
        np.testing.assert_array_equal(b0[k], b1[k], err_msg=k)
    g = np.load(os.path.join(GOLDEN, "protnote_small_concatenation.npz"))
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    s1 = s2 = 0.0
    n = 0
    for rows in ([0, 3, 5], [1, 2, 4]):
        lmax = int(lens[rows].max())
        f = O.masked_conv1d(x[rows][:, :, :lmax], lens[rows], sd["sequence_encoder.conv1.weight"],
                            sd["sequence_encoder.conv1.bias"], 1).double()
        s1 = s1 + f.sum(dim=(0, 2))
        s2 = s2 + (f * f).sum(dim=(0, 2))
        n += f.shape[0] * f.shape[2]
    assert n == 3 * 50 + 3 * 21
    mean = s1 / n
    var_unb = (s2 / n - mean * mean) * n / (n - 1)
    key, mom = "sequence_encoder.resnet_blocks.0.bn_activation_1.0.", 0.01
    np.testing.assert_allclose(b0[key + "running_mean"], (1 - mom) * sd[key + "running_mean"].numpy() + mom * mean.numpy(),
                               rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b0[key + "running_var"], (1 - mom) * sd[key + "running_var"].numpy() + mom * var_unb.numpy(),
                               rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("train_encoder,mode", [(False, "even"), (True, "even"), (True, "uneven_B")])
def test_sync_bn_two_ranks_equal_one_rank_on_the_whole_batch(train_encoder, mode):
    """SYNC_BN: True (bin/main.py:449-450).  Two ranks, three proteins each, statistics synchronised in the forward and
    the backward of every BatchNorm (encoder, W_p, W_l, output MLP incl. the closed-form first pair layer) must reproduce
    ONE rank running the six proteins (mode "uneven_B": four proteins on one rank and two on the other, each rank's loss
    weighted by its share - the BatchNorm row counts then differ per rank and must be summed, not multiplied by the world
    size): the mean of the two (weighted) losses is the whole-batch loss, the AVERAGED gradients equal the
    whole-batch gradients, and the BatchNorm running statistics agree - except W_l's running variances, whose label
    rows are replicated on every rank: SyncBatchNorm counts them world times, which only changes the unbiased-variance
    factor from N/(N-1) to 2N/(2N-1)."""
    res = _run_sync_bn(train_encoder, mode)
    (_, l0, g0, b0, full, g_ctl), (_, l1, g1, b1, _, _) = res
    l_full, g_full, b_full = full
    np.testing.assert_allclose(0.5 * (l0 + l1), l_full, rtol=2e-6)
    np.testing.assert_array_equal(g0, g1)                       # one all-reduce -> the same averaged gradient everywhere
    assert np.linalg.norm(g0) > 0
    rel = np.linalg.norm(g0 - g_full) / np.linalg.norm(g_full)
    assert rel < 2e-5, rel
    # control: with per-rank statistics (SYNC_BN: False) the same two ranks do NOT reproduce the whole-batch gradient
    assert np.linalg.norm(g_ctl - g_full) / np.linalg.norm(g_full) > 1e-2
    n_lab, mom = 10, 0.1
    g = np.load(os.path.join(GOLDEN, "protnote_small_concatenation.npz"))
    for k, v in b_full.items():
        if k.endswith("num_batches_tracked"):
            continue
        if k.startswith("W_l.") and k.endswith("running_var"):
            before = g["sd/" + k]
            ratio = (b0[k] - (1 - mom) * before) / (v - (1 - mom) * before)
            np.testing.assert_allclose(ratio, (2 * n_lab / (2 * n_lab - 1)) / (n_lab / (n_lab - 1)), rtol=1e-3, err_msg=k)
        else:
            np.testing.assert_allclose(b0[k], v, rtol=2e-5, atol=1e-6, err_msg=k)
            np.testing.assert_allclose(b1[k], v, rtol=2e-5, atol=1e-6, err_msg=k)
