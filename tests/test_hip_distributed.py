"""Two ranks on ONE MI355X (PN_SHARE_GPU=1, gloo moving the device tensors): the data-parallel code paths with real
HIP buffers - flat-gradient all-reduce inside train_step, BN-buffer broadcast, and the cross-rank evaluation metrics
(exact AP gathers the score columns of all ranks, binned AUPRC all-reduces its histograms).  On the 8-GPU node the same
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
    n, nl = 600, 37
    logits = torch.randn(n, nl, generator=g) * 2
    y = (torch.rand(n, nl, generator=g) < 0.1) | ((logits > 2) & (torch.rand(n, nl, generator=g) < 0.5))
    return torch.sigmoid(logits).float(), y


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PN_SHARE_GPU="1", PN_DIST_BACKEND="gloo")
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
    dev = "cuda:0"
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
    with torch.no_grad():  # make the ranks' BN buffers differ: the step must start from rank 0's
        for b in model.buffers():
            if b.is_floating_point():
                b.add_(0.01 * r)
    train_step(model, loss_fn, opt, batch, world_size=w)
    q.put((r, m_ap["map_micro"], m_ap["map_macro"], m_ap["ap_per_label"].cpu().numpy(), m_bn["map_micro"],
           m_bn["map_macro"], opt.flat_w.cpu().numpy()))
    torch.distributed.destroy_process_group()


def test_two_ranks_share_one_gpu():
    from oracle import metrics_oracle as MO

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
    bmacro = np.nanmean([MO.binned_auprc(p[:, j].numpy(), y[:, j].numpy(), thr) for j in range(p.shape[1])])
    for _, mi, ma, apl, bmi, bma, _w in res:  # every rank reports the metric of the WHOLE evaluation set
        np.testing.assert_allclose(mi, micro, rtol=1e-12)
        np.testing.assert_allclose(ma, np.nanmean(per), rtol=1e-12)
        np.testing.assert_allclose(apl, per, rtol=1e-12)
        np.testing.assert_allclose(bmi, bmicro, rtol=1e-12)
        np.testing.assert_allclose(bma, bmacro, rtol=1e-12)
    # averaged gradients + identical start => identical parameters on both ranks after the step
    np.testing.assert_array_equal(res[0][6], res[1][6])
