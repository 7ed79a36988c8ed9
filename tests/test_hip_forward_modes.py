"""Every (module mode, autograd mode) combination of ProtNote.forward - the reference restricts none of them
(protnote/models/ProtNote.py:168-334):

  train + autograd  the usual training step (tests/test_hip_train.py)
  train + no_grad   BatchNorm still takes batch statistics and advances its buffers, nothing to differentiate
  eval  + autograd  BatchNorm on its running statistics; logits (ensembled or not) are differentiable
  eval  + no_grad   the fused inference kernels (tests/test_hip_parity.py)

HIP path vs the CPU oracle (itself pinned to the reference's goldens) on the golden models' weights."""
import os

import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import replay_label_noise  # noqa: F401
from tests.helpers import make_protnote

pytestmark = pytest.mark.gpu
DEV = "cuda"
FUSIONS = ("concatenation", "concatenation_diff", "concatenation_prod", "similarity")


def _case(golden_dir, fusion):
    g = np.load(os.path.join(golden_dir, f"protnote_small_{fusion}.npz"))
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])
    cnt = torch.from_numpy(g["label_token_counts"])
    y = torch.from_numpy(g["multihots"])
    return g, x, lens, lab, cnt, y


@pytest.mark.parametrize("fusion", FUSIONS)
def test_train_mode_forward_under_no_grad(golden_dir, fusion, monkeypatch):
    """model.train() inside torch.no_grad(): logits of the train-mode arithmetic (batch-statistics BatchNorm over the
    pair grid, label noise), every BatchNorm buffer advanced exactly as a differentiable training forward advances it,
    num_batches_tracked counted."""
    g, x, lens, lab, cnt, y = _case(golden_dir, fusion)
    lab1, cnt1 = lab[0::2].contiguous(), cnt[0::2].contiguous()
    u = torch.from_numpy(g["train/noise_u"])
    model, sd = make_protnote(g, DEV)
    model.train()
    replay_label_noise(monkeypatch, lambda t, *a, **k: u.to(t.device).clone())
    with torch.no_grad():
        logits, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab1.to(DEV),
                          label_token_counts=cnt1.to(DEV))
    assert not logits.requires_grad
    ref = O.protnote_forward(sd, x, lens, lab1, fusion=fusion, training=True, noise_alpha=20.0, noise_u=u,
                             label_token_counts=cnt1, temperature=float(g["head_cfg_temperature"]))
    np.testing.assert_allclose(logits.cpu().numpy(), ref.numpy(), atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), g["train_BCE/logits"], atol=5e-4, rtol=1e-4)  # the reference's own run
    got = {k: v.cpu() for k, v in model.state_dict().items()}
    moved = 0
    for k in g.files:
        if k.startswith("train_BCE/sd_after/") and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            name = k[len("train_BCE/sd_after/"):]
            np.testing.assert_allclose(got[name].numpy(), g[k], atol=3e-5, rtol=2e-4, err_msg=name)
            moved += int(not np.array_equal(g[k], g["sd/" + name]))
    assert moved > 10


def test_no_grad_forward_leaves_a_pending_backward_intact(golden_dir):
    """A torch.no_grad() forward between a training forward and its backward (e.g. a quick validation probe in train
    mode): the probe runs on a temporary activation store, the pending backward still sees its own activations."""
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    g, x, lens, lab, cnt, y = _case(golden_dir, "concatenation")
    lab1 = lab[0::2].contiguous().to(DEV)
    xs, ls, yf = x.to(DEV), lens.to(DEV), y.float().to(DEV)
    grads = []
    for probe in (False, True):
        model, _ = make_protnote(g, DEV)
        model.label_embedding_noising_alpha = 0.0
        model.train()
        logits, _ = model(sequence_onehots=xs, sequence_lengths=ls, label_embeddings=lab1)
        if probe:
            with torch.no_grad():
                other, _ = model(sequence_onehots=xs.flip(0), sequence_lengths=ls.flip(0), label_embeddings=lab1[:5].contiguous())
            assert other.shape == (x.shape[0], 5)
        BCEWithLogitsLoss()(logits, yf).backward()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 20
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


@pytest.mark.parametrize("ndesc", (1, 2))
@pytest.mark.parametrize("fusion", FUSIONS)
def test_eval_mode_forward_is_differentiable(golden_dir, fusion, ndesc):
    """model.eval() with autograd on: logits equal the inference kernels' (BatchNorm on running statistics, no noise),
    d(loss)/d(every head parameter) equals the oracle's autograd through eval-mode BatchNorm - with and without the
    description ensembling logit(mean(sigmoid)) in the graph - and no BatchNorm buffer moves."""
    g, x, lens, lab, cnt, y = _case(golden_dir, fusion)
    T = float(g["head_cfg_temperature"])
    if ndesc == 1:
        lab, yy = lab[0::2].contiguous(), y
    else:
        yy = y
    model, sd = make_protnote(g, DEV)
    model.inference_descriptions_per_label = ndesc
    model.eval()
    for n_, p_ in model.named_parameters():
        if n_.startswith("sequence_encoder"):
            p_.requires_grad = False
    before = {k: v.clone() for k, v in model.state_dict().items()}
    xs, ls, labd = x.to(DEV), lens.to(DEV), lab.to(DEV)
    with torch.no_grad():
        fused, _ = model(sequence_onehots=xs, sequence_lengths=ls, label_embeddings=labd)
    logits, _ = model(sequence_onehots=xs, sequence_lengths=ls, label_embeddings=labd)
    assert logits.requires_grad and logits.shape == (x.shape[0], lab.shape[0] // ndesc)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), fused.cpu().numpy(), atol=2e-5, rtol=2e-5)
    key = "eval/logits_ens2" if ndesc == 2 else None
    if key:
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[key], atol=5e-4, rtol=1e-4)  # the reference's eval run
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, yy.float().to(DEV))
    loss.backward()
    # oracle: the same graph on CPU
    names = O.trainable_names(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    ref = O.protnote_forward(work, x, lens, lab, fusion=fusion, training=False, temperature=T, descriptions_per_label=ndesc)
    rl = torch.nn.functional.binary_cross_entropy_with_logits(ref, yy.float())
    rgrads = dict(zip(names, torch.autograd.grad(rl, [leaves[k] for k in names], allow_unused=True)))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref.detach().numpy(), atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(loss.item(), rl.item(), rtol=1e-4)
    named = dict(model.named_parameters())
    checked = 0
    for k, rg in rgrads.items():
        if rg is None:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k
            continue
        got = named[k].grad.cpu().numpy()
        np.testing.assert_allclose(got, rg.numpy(), atol=2e-5 + 2e-4 * float(rg.abs().max()), err_msg=k)
        checked += 1
    assert checked >= (14 if fusion == "similarity" else 20)
    after = model.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), k  # eval mode: no running statistic, no counter moved


def test_eval_mode_gradients_reach_the_inputs(golden_dir):
    """The reference hands label_embeddings / sequence_embeddings to the heads undetached (ProtNote.py:192-196,243-247):
    inputs that require grad get one, in either mode."""
    g, x, lens, lab, cnt, y = _case(golden_dir, "concatenation")
    model, sd = make_protnote(g, DEV)
    model.eval()
    for p_ in model.parameters():
        p_.requires_grad = False
    P_f = torch.from_numpy(g["eval/P_f"]).to(DEV).requires_grad_(True)
    labd = lab[0::2].contiguous().to(DEV).requires_grad_(True)
    logits, _ = model(sequence_embeddings=P_f, label_embeddings=labd)
    logits.square().mean().backward()
    Pc = torch.from_numpy(g["eval/P_f"]).clone().requires_grad_(True)
    Lc = lab[0::2].contiguous().clone().requires_grad_(True)
    ref = O.protnote_forward(sd, None, None, Lc, fusion="concatenation", training=False, sequence_embeddings=Pc)
    ref.square().mean().backward()
    np.testing.assert_allclose(P_f.grad.cpu().numpy(), Pc.grad.numpy(), atol=1e-6 + 2e-4 * float(Pc.grad.abs().max()))
    np.testing.assert_allclose(labd.grad.cpu().numpy(), Lc.grad.numpy(), atol=1e-6 + 2e-4 * float(Lc.grad.abs().max()))


def test_encoder_eval_mode_is_differentiable(golden_dir):
    """ProteInfer.get_embeddings in eval mode with trainable parameters (a second caller of the encoder, reference
    bin/test_proteinfer.py:303 / utils/main_utils.py:82, under autograd): embeddings of the inference kernels, gradients of
    the oracle's autograd through eval-mode BatchNorm, buffers untouched."""
    from tests.helpers import make_encoder, npz_cfg

    g = np.load(os.path.join(golden_dir, "encoder_small.npz"))
    sd = O.as_torch_sd(g, "sd/")
    enc = make_encoder(sd, "", npz_cfg(g, "cfg_"), DEV).eval()
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    before = {k: v.clone() for k, v in enc.state_dict().items()}
    emb = enc.get_embeddings(x.to(DEV), lens.to(DEV))
    assert emb.requires_grad
    np.testing.assert_allclose(emb.detach().cpu().numpy(), g["eval/embeddings"], atol=1e-4, rtol=1e-4)
    w = torch.linspace(-1, 1, emb.shape[1])
    (emb * w.to(DEV)).sum().backward()
    names = [k for k in sd if not k.startswith("output_layer") and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    ref = O.proteinfer_get_embeddings(work, x, lens, False)
    rg = dict(zip(names, torch.autograd.grad((ref * w).sum(), [leaves[k] for k in names])))
    named = dict(enc.named_parameters())
    for k, r in rg.items():
        np.testing.assert_allclose(named[k].grad.cpu().numpy(), r.numpy(), atol=1e-5 + 3e-4 * float(r.abs().max()), err_msg=k)
    for k, v in before.items():
        assert torch.equal(v, enc.state_dict()[k]), k
