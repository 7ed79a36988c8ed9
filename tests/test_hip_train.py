"""GPU parity tests of the training path (train-mode BatchNorm over the pair grid, backward, fused loss,
clip + Adam) against the reference-generated golden vectors and the CPU oracle."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import replay_label_noise  # noqa: F401
from tests.helpers import make_protnote, random_encoder_sd, random_head_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_gemm_tn_vs_torch():
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(1)
    for R, M, N in ((1000, 128, 128), (70001, 200, 52), (33, 4, 260), (5000, 3072, 128)):
        A = torch.randn(R, M, generator=g)
        Bm = torch.randn(R, N, generator=g) + torch.arange(N) * 0.01
        ref = A.double().T @ Bm.double()
        Ad, Bd = A.to(DEV), Bm.to(DEV)
        Cd = torch.full((M, N), float("nan"), device=DEV)
        ws = torch.empty(8 * M * N * 4 + 1024, dtype=torch.uint8, device=DEV)
        L.check(L.lib().pn_gemm_tn(L.ptr(Ad), M, L.ptr(Bd), N, L.ptr(Cd), N, R, M, N, L.ptr(ws), ws.numel(),
                                   L.stream_ptr()))
        torch.cuda.synchronize()
        err = (Cd.cpu().double() - ref).abs().max().item()
        assert err <= 3e-6 * ref.abs().max().item() * max(1.0, R ** 0.5 / 8), (R, M, N, err)


def test_fused_loss_and_metrics_golden(golden_dir):
    from protnote_amd.utils.losses import BCEWithLogitsLoss, FocalLoss
    from protnote_amd.models.ProtNoteTrainer import calculate_tp_fn_fp, calculate_f1, calculate_f1_micro

    g = _g(golden_dir, "losses_metrics.npz")
    logits = torch.from_numpy(g["logits"]).to(DEV)
    y = torch.from_numpy(g["multihots"]).to(DEV)
    cases = [("BCE", BCEWithLogitsLoss(pos_weight=float(g["BCE/pos_weight"]))),
             ("BCE_pw", BCEWithLogitsLoss(pos_weight=torch.tensor(float(g["BCE_pw/pos_weight"]))))]
    for name in ("Focal", "Focal_a", "Focal_ls"):
        gamma, alpha, ls = (float(v) for v in g[name + "/params"])
        cases.append((name, FocalLoss(alpha=alpha, gamma=gamma, label_smoothing=ls)))
    for name, fn in cases:
        for tgt in (y.float(), y):  # the trainer passes .float(); int64 multihots are accepted too
            lg = logits.clone().requires_grad_(True)
            l = fn(lg, tgt)
            l.backward()
            np.testing.assert_allclose(l.item(), float(g[name + "/loss"]), rtol=2e-6, err_msg=name)
            np.testing.assert_allclose(lg.grad.cpu().numpy(), g[name + "/dlogits"], atol=2e-10, rtol=2e-5,
                                       err_msg=name)
    # K13 + K14 fused: the loss pass itself accumulates the per-label counts
    fused = BCEWithLogitsLoss()
    fused.metric_counts = torch.zeros(3, logits.shape[1], device=DEV)
    fused.decision_threshold = 0.3
    fused(logits, y.float())
    fused(logits, y.float())
    for k, name in enumerate(("tp", "fn", "fp")):
        assert np.array_equal(fused.metric_counts[k].cpu().numpy(), 2 * g["th0.3/" + name]), name
    for th in (0.5, 0.3):
        tp, fn_, fp = calculate_tp_fn_fp(torch.sigmoid(logits), y, threshold=th)
        assert np.array_equal(tp.cpu().numpy(), g[f"th{th}/tp"])  # integer counts: bit-exact
        assert np.array_equal(fn_.cpu().numpy(), g[f"th{th}/fn"])
        assert np.array_equal(fp.cpu().numpy(), g[f"th{th}/fp"])
        np.testing.assert_allclose(calculate_f1(tp, fn_, fp).cpu().numpy(), g[f"th{th}/f1"], rtol=1e-6)
        np.testing.assert_allclose(calculate_f1_micro(tp, fn_, fp).cpu().numpy(), g[f"th{th}/f1_micro"], rtol=1e-6)


def _assert_adam_close(got, ref, name, lr=3e-4, steps=1, atol=3e-5, rtol=2e-4, frac=0.995):
    """Post-Adam parameters.  Adam's update lr*m/(sqrt(v)+eps) is sign-like for gradients at f32-noise level
    (|g| ~ 1e-8): such an element moves by +-lr per step whichever way the last-bit noise points (CPU vs GPU
    summation order; the GPU itself is bit-reproducible, see test_train_step_bit_reproducible).  Hence: every element
    within the worst case 2*lr*steps, and at least `frac` of them within (atol, rtol)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    diff = np.abs(got - ref)
    assert diff.max() <= 2 * lr * steps + atol, (name, diff.max())
    ok = diff <= atol + rtol * np.abs(ref)
    assert ok.mean() >= frac, (name, ok.mean())


def _freeze_encoder(model):
    for n, p in model.named_parameters():
        if n.startswith("sequence_encoder"):
            p.requires_grad = False


@pytest.mark.parametrize("fusion", ["concatenation", "concatenation_diff", "concatenation_prod", "similarity"])
@pytest.mark.parametrize("loss", ["BCE", "FocalLoss"])
def test_train_step_golden(golden_dir, fusion, loss, monkeypatch):
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam
    from protnote_amd.models.train_path import head_parameters

    g = _g(golden_dir, f"protnote_small_{fusion}.npz")
    model, _ = make_protnote(g, DEV)
    _freeze_encoder(model)
    model.train()
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV)
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous().to(DEV)
    y = torch.from_numpy(g["multihots"]).to(DEV)
    u = torch.from_numpy(g["train/noise_u"]).to(DEV)
    replay_label_noise(monkeypatch, lambda t, *a, **k: u.clone())
    cfg = {"params": {"LOSS_FN": loss, "FOCAL_LOSS_GAMMA": 2, "FOCAL_LOSS_ALPHA": -1, "LABEL_SMOOTHING": 0.0}}
    loss_fn = get_loss(cfg, bce_pos_weight=torch.tensor(1.0))
    opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
    logits, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab, label_token_counts=cnt)
    l = loss_fn(logits, y.float())
    l.backward()
    p = f"train_{loss}/"
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[p + "logits"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(l.item(), float(g[p + "loss"]), rtol=1e-4)
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith(p + "grad/"):
            name = k[len(p + "grad/"):]
            ref = g[k]
            got = named[name].grad.cpu().numpy()
            np.testing.assert_allclose(got, ref, atol=2e-5 + 2e-4 * np.abs(ref).max(), err_msg=name)
    opt.step()
    np.testing.assert_allclose(opt.last_grad_norm.item(), float(g[p + "grad_norm"]), rtol=2e-4)
    got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k in g.files:
        if k.startswith(p + "sd_after/"):
            name = k[len(p + "sd_after/"):]
            if name.endswith(("running_mean", "running_var", "num_batches_tracked")) or name.startswith("sequence_encoder"):
                np.testing.assert_allclose(got[name], g[k], atol=3e-5, rtol=2e-4, err_msg=name)
            else:
                _assert_adam_close(got[name], g[k], name)


@pytest.mark.parametrize("B,NL,chunk", [(8, 40, 3), (130, 40, 7), (8, 9, None), (100, 660, 256), (256, 512, 100),
                                        (8, 16400, None), (300, 24, 5)])
def test_train_real_width_vs_oracle(B, NL, chunk):
    """d=1024 / h=3072 / 3 hidden layers / 4-layer projection heads: logits, loss and every gradient of one
    train-mode step against the oracle's autograd on the same seeded inputs.  The 100 x 660 grid (66 000 pair rows,
    odd batch, ragged last tile, three backward chunks) is the smallest that runs on the 256-tile LDS-DMA kernels
    (forward NT, chunked dh NT, TN weight gradients), so those are held to the reference algorithm directly.
    256 x 512 is the bench's per-GPU batch (BASELINE configs[2]: B = 256): the scalar pair decode of the specialised
    weight-gradient kernel (B % 32 == 0, whole-slab splits), the top layer's dh written in six label chunks over its
    own consumed rows (the last one ragged), 131 072 pair rows.  8 x 16 400 takes the label table past 16 384 rows:
    the W_l backward on the materialised-dY path (256-tile NT / big TN kernels) - the two regimes of the full-size step
    that were only ever compared with themselves.  (f64 oracle: ~35 GB of host memory, ~1 min on the GPU box's host.)
    300 x 24: more than 256 proteins - the layer-1 backward takes its two-pass reductions (k_pair_mask_reduce<0|1>)
    instead of the single-pass kernel every B <= 256 grid above runs (k_pair_mask_reduce_fused)."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(21)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()

    def oracle(dtype, where="cpu"):
        ref_sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()).to(where) for k, v in sd.items()}
        names = O.trainable_names(ref_sd)
        leaves = {k: ref_sd[k].clone().requires_grad_(True) for k in names}
        work = dict(ref_sd)
        work.update(leaves)
        lg = O.protnote_forward(work, None, None, lab.to(where, dtype), training=True,
                                sequence_embeddings=P_f.to(where, dtype))
        ls = O.bce_loss(lg, y.to(where, dtype))
        grads = dict(zip(names, (g_.cpu() for g_ in torch.autograd.grad(ls, [leaves[k] for k in names]))))
        return lg.detach().cpu(), ls.detach().cpu(), grads, {k: v.detach().cpu() for k, v in work.items()}

    # ground truth in f64 - the oracle's naive formulation (joint tensor, autograd) in stock torch ops, evaluated on the
    # device for the big grids (float64 there is the same ground truth to ~1e-13 and takes seconds instead of a minute of
    # host time); the reference's own f32 CPU path gives the error scale to hold the GPU to and stays on the CPU
    ref_logits, ref_loss, ref_grads, work = oracle(torch.float64, DEV if B * NL >= 50000 else "cpu")
    _, _, cpu32_grads, _ = oracle(torch.float32)
    torch.cuda.empty_cache()

    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    model.pair_label_chunk = chunk
    logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    l = BCEWithLogitsLoss()(logits, y.to(DEV))
    l.backward()
    assert (logits.detach().cpu().double() - ref_logits).abs().max().item() < 5e-4
    np.testing.assert_allclose(l.item(), ref_loss.item(), rtol=1e-4)
    bad = []
    for name, p in model.named_parameters():
        ref = ref_grads[name]
        # ReLU masks of pre-activations within f32 rounding of 0 flip under any change of summation order (each
        # flip is a rank-1 O(dl*h) change) and every gradient inherits the ~1e-4 abs error of the f32 logits
        # through dl = sigmoid(x) - y, so element-wise 1e-4 is not meaningful for gradients on large pair grids
        # (the reference's own f32 CPU path is 5e-4..2e-3 away from f64 at B >= 128; printed by this test under -s).
        # Require the GPU's Frobenius error vs f64 to be of the same class as the f32 CPU path's.
        nrm = max(ref.norm().item(), 1e-30)
        rel = (p.grad.cpu().double() - ref).norm().item() / nrm
        rel_cpu = (cpu32_grads[name].double() - ref).norm().item() / nrm
        # Measured (round 2, `pytest -s` prints): the GPU's error is 1.1x..2.9x the f32 CPU path's on every tensor and
        # grid (e.g. 130x40: 1.5e-3 vs 8.4e-4; 8x9: 4.1e-6 vs 1.5e-6), deterministically - hence the factor 4.
        print(f"grad-err {B}x{NL} {name}: gpu {rel:.2e} cpu32 {rel_cpu:.2e} ratio {rel / max(rel_cpu, 1e-30):.2f}")
        # 256 x 512 (round 3, measured): every tensor 1.4x..1.9x EXCEPT W_l.0 .. W_l.8 at 3.6x..4.2x (3.0e-3 vs 7.5e-4).
        # With B = 256 proteins per label the label-side gradient dL_e is dominated by its common mode (the mean over
        # labels grows like B, the per-label signal like sqrt(B): ratio ~ 0.67 sqrt(B) = 11), which W_l's last BatchNorm
        # backward subtracts (du - mean(du)): whatever element-wise f32 noise dL_e carries is amplified ~10x behind that
        # BatchNorm while a noise component common to all labels cancels - W_l.12 / W_l.9 in front of it sit at 1.5x / 1.9x
        # like everything else, and 100 x 660 / 8 x 16400 (smaller B) show no such step.  Still the f32 class (< 4e-3);
        # the factor for those tensors (and W_l.9.bias, the same BatchNorm's d beta = sum_j du: 4.4x) on that grid is 6.
        behind = name.startswith("W_l.") and (int(name.split(".")[1]) <= 8 or name == "W_l.9.bias")
        factor = 6 if (B >= 256 and behind) else 4
        if not (rel < max(factor * rel_cpu, 1e-6) and rel < 4e-3):
            bad.append((name, rel, rel_cpu))
    assert not bad, bad
    # BN running statistics after the train-mode forward
    got = {k: v.cpu() for k, v in model.state_dict().items()}
    for k, v in work.items():
        if k.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(got[k].numpy(), v.detach().float().numpy(), atol=1e-5, rtol=1e-4, err_msg=k)


def test_config0_shape_one_epoch_vs_oracle(golden_dir):
    """BASELINE configs[0] shape: 64 synthetic sequences (L <= 128), 256 labels, batch 16, one epoch = 4
    optimisation steps with label noise, BCE, clip 1, Adam 3e-4 - the object graph bin/main.py builds
    (ProteInfer -> ProtNote -> get_loss -> train-step body), full-width model.  HIP vs the CPU oracle after the
    whole epoch: loss trajectory, parameters, BN buffers of the (train-mode) frozen encoder - AND vs the REFERENCE itself
    run on the same seeded case at this width (tests/golden/config0_full_width.npz, generated by make_golden.py)."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer
    from protnote_amd.models.ProtNoteTrainer import train_step
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam
    from tests.helpers import config0_case

    c = config0_case()
    ecfg, sd, lab, cnt, y_all, noises, batch = c["ecfg"], c["sd"], c["lab"], c["cnt"], c["y_all"], c["noises"], c["batch"]
    NSEQ, NL, BS = c["n_steps"] * c["BS"], c["NL"], c["BS"]

    # ---- oracle epoch ----
    osd = {k: v.clone() for k, v in sd.items()}
    st = {}
    ref_losses = []
    for k in range(NSEQ // BS):
        x, lens, y = batch(k)
        _, l, _, _ = O.train_step(osd, x, lens, lab, y, loss="BCE", noise_alpha=20.0, noise_u=noises[k],
                                  label_token_counts=cnt, adam_state=st)
        ref_losses.append(float(l))

    # ---- HIP epoch ----
    enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
    model = ProtNote(sequence_encoder=enc, output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3,
                     projection_head_num_layers=4, projection_head_hidden_dim_scale_factor=3,
                     label_embedding_noising_alpha=20.0)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    _freeze_encoder(model)
    opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    counts = torch.zeros(3, NL, device=DEV)
    real = torch.rand_like
    model.label_noise_rng = "torch"  # replay the reference run's noise through torch.rand_like
    losses = []
    try:
        for k in range(NSEQ // BS):
            x, lens, y = batch(k)
            u = noises[k].to(DEV)
            torch.rand_like = lambda t, *a, **kw: u.clone()
            b = {"sequence_onehots": x.to(DEV), "sequence_lengths": lens.to(DEV), "label_embeddings": lab.to(DEV),
                 "label_token_counts": cnt.to(DEV), "label_multihots": y.to(DEV)}
            losses.append(float(train_step(model, loss_fn, opt, b, counts=counts)))
    finally:
        torch.rand_like = real
    # tiny-batch BN (16 rows) + Adam amplify f32 rounding step over step: 1e-6 after step 1, ~1e-3 after 4
    np.testing.assert_allclose(losses[:2], ref_losses[:2], rtol=1e-4)
    np.testing.assert_allclose(losses, ref_losses, rtol=6e-3)
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for k, v in osd.items():
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v), k
        elif k.startswith("sequence_encoder"):
            # the frozen encoder's train-mode BN buffers do not depend on the trained weights: tight
            np.testing.assert_allclose(got[k].numpy(), v.numpy(), atol=2e-6, rtol=1e-5, err_msg=k)
        elif k.endswith(("running_mean", "running_var")):
            # head BN buffers see the +-lr noise-direction moves of W (below) amplified by large-mean input
            # features (dW[n,k] = sum_i dY[i,n] X[i,k] with sum_i dY = 0 after BatchNorm: for near-constant
            # columns X[:,k] the gradient is rounding noise, Adam turns it into +-lr, and |mean X[:,k]| ~ 10 turns
            # that into 1e-2 shifts of the batch mean - which the BatchNorm then removes again; measured with
            # measured in round 1).  Only the bulk is comparable.
            d = (got[k] - v).abs()
            assert d.mean().item() < 2e-3 * max(v.abs().mean().item(), 1.0), k
        else:
            # Adam's update lr*m/(sqrt(v)+eps) is sign-like for gradients at f32-noise level (|g| ~ eps): such an
            # element moves by +-lr per step whichever way the noise points, on CPU and GPU independently.  So:
            # every element within the 4-step worst case 2*lr*4, and >= 95 % within 5 % of ONE step.
            # (see _assert_adam_close) after 4 steps: hard bound 2*lr*4 on every element, and the mean deviation
            # below 25 % of the 4-step Adam travel lr*4
            d = (got[k] - v).abs()
            assert d.max().item() <= 2 * 3e-4 * 4 + 1e-5, (k, d.max().item())
            assert d.mean().item() <= 0.25 * 3e-4 * 4, (k, d.mean().item())
    assert float(counts.sum()) > 0 and float(counts[0].sum() + counts[1].sum()) == float(y_all.sum())
    # ---- the same epoch as the REFERENCE ran it (real width; reference-generated vectors) ----
    g = _g(golden_dir, "config0_full_width.npz")
    fresh = ProtNote(sequence_encoder=ProteInfer(activation=torch.nn.ReLU, **ecfg), output_mlp_hidden_dim_scale_factor=3,
                     output_mlp_num_layers=3, projection_head_num_layers=4, projection_head_hidden_dim_scale_factor=3,
                     inference_descriptions_per_label=2)
    fresh.load_state_dict(sd)
    fresh = fresh.to(DEV).eval()
    x0, lens0, _ = batch(0)
    with torch.no_grad():  # inference at the real width before any training, 2 descriptions per label ensembled
        ens, _ = fresh(sequence_onehots=x0.to(DEV), sequence_lengths=lens0.to(DEV), label_embeddings=lab.to(DEV))
        pf = fresh.sequence_encoder.get_embeddings(x0.to(DEV), lens0.to(DEV))
    np.testing.assert_allclose(pf.cpu().numpy(), g["eval0/P_f"], atol=1e-4, rtol=1e-4)
    # (the random full-width weights give raw logits of +-15 and more: where the ensembled probability is within 1e-6 of 0 or
    #  1, torch.special.logit(eps=1e-7) has a slope of 1e6..1e7 and f32 rounding of the sigmoid mean moves it by 1e-2 - held
    #  to 5e-4 inside |logit| < 12 and to 0.05 in the saturated tail, ProtNote.py:313-322)
    fresh.inference_descriptions_per_label = 1
    with torch.no_grad():
        raw, _ = fresh(sequence_onehots=x0.to(DEV), sequence_lengths=lens0.to(DEV), label_embeddings=lab.to(DEV))
    np.testing.assert_allclose(raw.cpu().numpy(), g["eval0/logits_raw"], atol=5e-4, rtol=1e-4)
    # the ensembled output is logit(mean(sigmoid)): with these random full-width weights almost every probability is within
    # 1e-5 of 0 or 1, where torch.special.logit(eps=1e-7) has a slope of 1e5..1e7 - compared as probabilities (1e-6) and, in
    # logit space, to 0.05 (ProtNote.py:313-322)
    e_ref, e_got = g["eval0/logits_ens2"], ens.cpu().numpy()
    np.testing.assert_allclose(1 / (1 + np.exp(-e_got.astype(np.float64))), 1 / (1 + np.exp(-e_ref.astype(np.float64))), atol=1e-6)
    np.testing.assert_allclose(e_got, e_ref, atol=0.05)
    n_par = 0
    for key in g.files:
        if key.startswith("after/buffer/"):
            name = key[len("after/buffer/"):]
            if name.endswith("num_batches_tracked"):
                assert int(got[name]) == int(g[key]), name
            elif name.startswith("sequence_encoder"):
                np.testing.assert_allclose(got[name].numpy(), g[key], atol=2e-6, rtol=1e-5, err_msg=name)
            else:
                d = np.abs(got[name].numpy() - g[key])
                assert d.mean() < 2e-3 * max(np.abs(g[key]).mean(), 1.0), name
        elif key.startswith("after/param_head/"):
            name = key[len("after/param_head/"):]
            d = np.abs(got[name].reshape(-1)[:256].numpy() - g[key])
            assert d.max() <= 2 * 3e-4 * 4 + 1e-5 and d.mean() <= 0.25 * 3e-4 * 4, (name, d.max(), d.mean())
            n_par += 1
    assert n_par == 31


def test_trainer_learns_synthetic_task():
    """End-to-end sanity of the epoch driver on a learnable toy task (small model): training loss falls, and the
    evaluation mAP on the training distribution rises well above the label prevalence."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer
    from protnote_amd.models.ProtNoteTrainer import Trainer
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam

    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(0)
    enc = ProteInfer(4, 20, 32, 9, torch.nn.ReLU, 3, 2, 0.5)
    model = ProtNote(protein_embedding_dim=32, label_embedding_dim=16, latent_dim=16, sequence_encoder=enc,
                     output_mlp_hidden_dim_scale_factor=2, output_mlp_num_layers=2, projection_head_num_layers=2,
                     projection_head_hidden_dim_scale_factor=2).to(DEV)
    for n, p in model.named_parameters():
        if n.startswith("sequence_encoder"):
            p.requires_grad = False
    NL, B, L = 12, 32, 40
    lab = torch.randn(NL, 16, generator=gen)
    motifs = torch.randint(0, 20, (NL, 6), generator=gen)  # label j <=> the sequence contains motif j

    def make_batch(seed):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, 20, (B, L), generator=g)
        y = (torch.rand(B, NL, generator=g) < 0.25).long()
        for b in range(B):
            for j in torch.nonzero(y[b]).flatten().tolist():
                pos = int(torch.randint(0, L - 6, (1,), generator=g))
                ids[b, pos:pos + 6] = motifs[j]
        x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
        return {"sequence_onehots": x.to(DEV), "sequence_lengths": torch.full((B,), L, device=DEV),
                "label_embeddings": lab.to(DEV), "label_multihots": y.to(DEV)}

    batches = [make_batch(s) for s in range(8)]
    # the "frozen" encoder normalises with batch statistics in train mode and with its running buffers in eval mode
    # (SURVEY 3.4-1); a pretrained encoder's buffers match its data, a random one's do not - calibrate them first
    # (momentum 0.01: 600 passes leave 0.2 % of the initial 0/1 buffers)
    model.train()
    with torch.no_grad():
        for k in range(600):
            b = batches[k % len(batches)]
            model.sequence_encoder.get_embeddings(b["sequence_onehots"], b["sequence_lengths"])
    opt = FusedClipAdam(head_parameters(model), lr=3e-3, max_norm=1.0)
    tr = Trainer(model, get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0)), opt)
    first = tr.train_one_epoch(batches)
    for _ in range(14):
        last = tr.train_one_epoch(batches)
    assert last["loss"] < 0.7 * first["loss"], (first, last)
    ev = tr.evaluate(batches)
    # label prevalence (= the AP of a random scorer) is 0.25; round-1 runs (then with atomics-induced trajectory noise)
    # gave map_micro 0.45..0.62 and f1_micro 0.43..0.52.  The run is bit-reproducible since round 2.
    assert ev["map_micro"] > 0.36 and ev["f1_micro"] > 0.3, ev
    # only_represented_labels (ProtNoteTrainer.py:469-472,517-519): metrics over a label subset = metrics of the
    # evaluation restricted to those columns
    nl = batches[0]["label_multihots"].shape[1]
    mask = torch.zeros(nl, dtype=torch.bool)
    mask[::2] = True
    sub = tr.evaluate(batches, represented_label_mask=mask)
    half = [dict(b, label_multihots=b["label_multihots"][:, ::2].contiguous(),
                 label_embeddings=b["label_embeddings"][::2].contiguous()) for b in batches]
    ref = tr.evaluate(half)
    assert abs(sub["map_macro"] - ref["map_macro"]) < 1e-6 and abs(sub["f1_micro"] - ref["f1_micro"]) < 1e-6, (sub, ref)


def test_torch_ddp_wrapper_compat(golden_dir):
    """The reference wraps the model in torch DDP (bin/main.py:452, find_unused_parameters=True) and runs
    scaler.scale(loss).backward() under autocast (ProtNoteTrainer.py:728-738).  With a 1-rank process group the
    same wrapper + GradScaler + autocast around protnote_amd's modules must give the plain gradients."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    from protnote_amd.utils.losses import get_loss

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g = _g(golden_dir, "protnote_small_concatenation.npz")
        model, _ = make_protnote(g, DEV)
        _freeze_encoder(model)
        model.label_embedding_noising_alpha = 0.0
        model.train()
        ddp = DDP(model, device_ids=[0], find_unused_parameters=True)
        x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
        lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV)
        y = torch.from_numpy(g["multihots"]).to(DEV)
        loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
        scaler = torch.amp.GradScaler("cuda")
        with torch.autocast("cuda"):
            logits, _ = ddp(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab)
            loss = loss_fn(logits, y.float())
        scaler.scale(loss).backward()
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=3e-4)
        scaler.unscale_(opt)
        # reference gradient of the same (noise-free) step from the oracle
        sd = O.as_torch_sd(g, "sd/")
        _, ref_loss, ref_grads, _ = O.train_step(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]), lab.cpu(),
                                                 y.cpu(), loss="BCE", apply_update=False)
        np.testing.assert_allclose(loss.item(), float(ref_loss), rtol=1e-4)
        named = dict(model.named_parameters())
        for k, ref in ref_grads.items():
            got = named[k].grad.cpu()
            assert (got - ref).abs().max().item() <= 2e-5 + 2e-4 * ref.abs().max().item(), k
    finally:
        dist.destroy_process_group()


def test_train_sequence_encoder_golden(golden_dir, monkeypatch):
    """TRAIN_SEQUENCE_ENCODER: True (reference ProtNote.py:248-256): gradients of every encoder parameter (conv
    weights/biases, BN affine) and of the heads vs the reference's autograd on the golden batch."""
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    g = _g(golden_dir, "protnote_small_concatenation.npz")
    model, _ = make_protnote(g, DEV, train_sequence_encoder=True)
    model.train()
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV)
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous().to(DEV)
    u = torch.from_numpy(g["train/noise_u"]).to(DEV)
    replay_label_noise(monkeypatch, lambda t, *a, **k: u.clone())
    logits, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab, label_token_counts=cnt)
    loss = BCEWithLogitsLoss()(logits, torch.from_numpy(g["multihots"]).to(DEV).float())
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["train_enc_BCE/loss"]), rtol=1e-4)
    named = dict(model.named_parameters())
    n_enc = 0
    for k in g.files:
        if k.startswith("train_enc_BCE/grad/"):
            name = k[len("train_enc_BCE/grad/"):]
            ref = g[k]
            got = named[name].grad.cpu().numpy()
            np.testing.assert_allclose(got, ref, atol=2e-5 + 3e-4 * np.abs(ref).max(), err_msg=name)
            n_enc += name.startswith("sequence_encoder.")
    assert n_enc == 18
    assert named["sequence_encoder.output_layer.weight"].grad is None  # never reached by get_embeddings


def _encoder_grad_errors(C, lens, seed):
    """TRAIN_SEQUENCE_ENCODER step (k = 9, 5 blocks, dilations 1..81, ragged lengths) on the HIP path vs the oracle's autograd
    in f64 (on the device) and in f32 on the CPU (the reference's arithmetic): per tensor (HIP error, CPU-f32 error, floor),
    Frobenius, relative to the f64 gradient's norm."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(seed)
    ecfg = dict(num_labels=8, input_channels=20, output_channels=C, kernel_size=9, dilation_base=3,
                num_resnet_blocks=5, bottleneck_factor=0.5)
    sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
    sd.update(random_head_sd(gen, C, 1024, 64, 128, 2, 128, 2))
    lens = torch.tensor(lens)
    B, Lmax, NL = len(lens), int(lens.max()), 24
    ids = torch.randint(0, 20, (B, Lmax), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.3).float()
    where = DEV if C >= 512 else "cpu"  # float64 ground truth: on the device for the wide model (seconds instead of a minute)
    osd = {k: (v.clone().double() if v.is_floating_point() else v.clone()).to(where) for k, v in sd.items()}
    _, l64, g64, _ = O.train_step(osd, x.double().to(where), lens.to(where), lab.double().to(where), y.double().to(where),
                                  loss="BCE", apply_update=False, train_sequence_encoder=True)
    l64, g64 = l64.cpu(), {k: v.cpu() for k, v in g64.items()}
    # the reference algorithm's own f32 CPU run gives the error scale (as in test_train_real_width_vs_oracle)
    _, _, g32, _ = O.train_step({k: v.clone() for k, v in sd.items()}, x, lens, lab, y, loss="BCE", apply_update=False,
                                train_sequence_encoder=True)
    enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
    model = ProtNote(protein_embedding_dim=C, sequence_encoder=enc, latent_dim=64,
                     output_mlp_hidden_dim_scale_factor=2, output_mlp_num_layers=2, projection_head_num_layers=2,
                     projection_head_hidden_dim_scale_factor=2, train_sequence_encoder=True)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    logits, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab.to(DEV))
    loss = BCEWithLogitsLoss()(logits, y.to(DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(l64), rtol=1e-4)
    gmax = max(g.abs().max().item() for n, g in g64.items() if n.startswith("sequence_encoder."))
    out = {}
    for name, p in model.named_parameters():
        if name.startswith("sequence_encoder.output_layer"):
            continue
        ref = g64[name]
        nrm = max(ref.norm().item(), 1e-30)
        # absolute floor: the last conv bias only shifts every P_f row equally, which W_p's BatchNorm removes - its
        # true gradient is exactly 0 and the f32 result is rounding noise
        floor = 3e-6 * gmax * ref.numel() ** 0.5 / nrm
        out[name] = ((p.grad.cpu().double() - ref).norm().item() / nrm, (g32[name].double() - ref).norm().item() / nrm, floor)
    return out


def test_train_sequence_encoder_narrow_vs_oracle():
    """52 channels: every gradient of the trainable encoder agrees with the f64 oracle to ~4e-6 (tight bound: the
    position / dilation geometry - lengths 1, < 4 x dilation, the full pad length - is exact)."""
    for name, (err, err32, floor) in _encoder_grad_errors(52, [300, 37, 1, 222, 300, 150], 41).items():
        assert err <= 2e-4 + floor, (name, err, err32)


def test_train_sequence_encoder_wide_vs_oracle():
    """The reference width (1100 / 550 channels): each of the 10 BatchNorm+ReLU layers has 3e5 .. 2e6 pre-activations, and
    the gradient is DISCONTINUOUS in their signs: one mask that differs from the f64 ground truth is a 5e-4 .. 5e-3 relative
    step in conv1's gradient (tools/relu_flip_probe.py; larger the smaller the batch).  Rounds 1-3 ran the forward
    convolutions as one k-ordered f32-MFMA chain over K = 9 x 1100 products (~3e-6 relative pre-activation error, ~50 flips):
    3e-3 .. 5e-3 in EVERY configuration, held to a 1e-2 cap - the class of stock torch / MIOpen f32 on this GPU (3.1e-3), not
    of the reference's f32 CPU run (2.7e-6).  Since round 4 the forward of a TRAINABLE encoder accumulates its wide
    convolutions in float64 on the matrix cores and rounds once (gemm_conv_f64.hpp): 1e-5 .. 2.5e-5 (tools/
    encoder_grad_error.py, profiles/r04_encoder_grad_error.json), what is left being the f32 chains of the linear, flip-free
    backward convolutions.  What no f32 implementation can exclude - the CPU's included - is the isolated pre-activation that
    sits within ONE rounding of zero (expected ~1 among 1e7): it shows as a single flip-sized step.  Measured on the three
    seeded configurations below: seed 41 - HIP 1.9e-5, CPU f32 2.7e-6; seed 43 - HIP 4.5e-4, CPU f32 4.3e-4 (the reference's
    own arithmetic flips a mask there); seed 42 - HIP 4.3e-3 (one flip on a 301-residue batch), CPU f32 2.9e-5.  Hence: every
    tensor of every configuration inside the old 1e-2 cap, and the CPU class - 1e-4, or twice the CPU-f32 error where that is
    larger - in at least two of the three.  (B >= 4: with B = 2 the batch-statistics BatchNorm in W_p is singular.)"""
    tight = 0
    for seed, lens in ((41, [300, 37, 1, 222, 300, 150]), (42, [100, 37, 64, 100]), (43, [300, 37, 1, 222, 300, 150])):
        errs = _encoder_grad_errors(1100, lens, seed)
        worst = max(errs.items(), key=lambda kv: kv[1][0] - kv[1][2])
        print(f"enc-grad-err C=1100 seed {seed}: conv1.weight gpu {errs['sequence_encoder.conv1.weight'][0]:.2e} "
              f"cpu32 {errs['sequence_encoder.conv1.weight'][1]:.2e}; worst {worst[0]} {worst[1][0]:.2e}")
        for name, (err, err32, floor) in errs.items():
            assert err <= 1e-2 + floor, (seed, name, err, err32)
        # CPU class: 1e-4 (100 x tighter than the old cap), or - where the reference's own f32 run has a flipped mask too
        # (seed 43: the CPU is at 4.3e-4 itself) - twice the CPU-f32 error
        tight += all(err <= max(1e-4, 2 * err32) + floor for err, err32, floor in errs.values())
    assert tight >= 2, tight


def test_gradient_accumulation_matches_single_step(golden_dir):
    """GRADIENT_ACCUMULATION_STEPS = 2 (ProtNoteTrainer.py:732-755): two backward passes of the same batch with the
    loss halved accumulate to the gradient of one plain step (no optimizer update in between), so the Adam update
    after the second pass equals the single-step update; nothing is applied after the first pass."""
    from protnote_amd.models.ProtNoteTrainer import train_step
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam

    g = _g(golden_dir, "protnote_small_concatenation.npz")

    def fresh():
        m, _ = make_protnote(g, DEV)
        _freeze_encoder(m)
        m.train()
        return m, FusedClipAdam(head_parameters(m), lr=3e-4, max_norm=1.0)

    batch = {"sequence_onehots": torch.from_numpy(g["x"]).to(DEV), "sequence_lengths": torch.from_numpy(g["lens"]).to(DEV),
             "label_embeddings": torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV),
             "label_multihots": torch.from_numpy(g["multihots"]).float().to(DEV)}
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    m1, o1 = fresh()
    l_full = train_step(m1, loss_fn, o1, batch)
    m2, o2 = fresh()
    w0 = o2.flat_w.clone()
    l_half = train_step(m2, loss_fn, o2, batch, gradient_accumulation_steps=2, batch_idx=0)
    assert torch.equal(o2.flat_w, w0) and o2.flat_g.abs().sum() > 0      # accumulated, not applied
    np.testing.assert_allclose(2 * float(l_half), float(l_full), rtol=1e-6)
    train_step(m2, loss_fn, o2, batch, gradient_accumulation_steps=2, batch_idx=1)
    assert o2.flat_g.abs().sum() == 0                                     # zero_grad after the update
    _assert_adam_close(o2.flat_w.cpu().numpy(), o1.flat_w.cpu().numpy(), "flat_w")


def test_build_training_from_config(golden_dir):
    """params -> (loss, fused optimiser, epoch driver): AdamW weight decay, CLIP_VALUE null, accumulation and the
    decision threshold reach the objects that use them; one epoch over two batches runs and updates the weights once
    (GRADIENT_ACCUMULATION_STEPS = 2)."""
    from protnote_amd.utils.configs import build_training

    g = _g(golden_dir, "protnote_small_concatenation.npz")
    model, _ = make_protnote(g, DEV)
    _freeze_encoder(model)
    cfg = {"params": {"LOSS_FN": "FocalLoss", "FOCAL_LOSS_GAMMA": 2, "FOCAL_LOSS_ALPHA": -1, "LABEL_SMOOTHING": 0.0,
                      "BCE_POS_WEIGHT": 1, "OPTIMIZER": "AdamW", "WEIGHT_DECAY": 0.001, "LEARNING_RATE": 1e-3,
                      "CLIP_VALUE": None, "GRADIENT_ACCUMULATION_STEPS": 2, "DECISION_TH": 0.3,
                      "TRAIN_SEQUENCE_ENCODER": False}}
    loss_fn, opt, trainer = build_training(cfg, model)
    assert opt.weight_decay == 0.001 and opt.max_norm is None and opt.lr == 1e-3
    assert trainer.threshold == 0.3 and trainer.gradient_accumulation_steps == 2
    batch = {"sequence_onehots": torch.from_numpy(g["x"]).to(DEV), "sequence_lengths": torch.from_numpy(g["lens"]).to(DEV),
             "label_embeddings": torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV),
             "label_multihots": torch.from_numpy(g["multihots"]).float().to(DEV)}
    w0 = opt.flat_w.clone()
    out = trainer.train_one_epoch([batch, batch])
    assert opt.step_count == 1 and not torch.equal(opt.flat_w, w0)
    assert np.isfinite(out["loss"]) and 0.0 <= out["f1_micro"] <= 1.0
    ev = trainer.evaluate([batch], estimate_map=True)
    assert 0.0 <= ev["map_micro"] <= 1.0


@pytest.mark.parametrize("loss", ["BCE", "FocalLoss"])
def test_output_mlp_without_batchnorm_golden(golden_dir, loss, monkeypatch):
    """OUTPUT_MLP_BATCHNORM: False - eval logits, train-step logits / loss / every gradient (Linear biases included)
    and the Adam update against the reference golden."""
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam
    from protnote_amd.models.train_path import head_parameters

    g = _g(golden_dir, "protnote_small_concatenation_nobn.npz")
    model, _ = make_protnote(g, DEV)
    assert not any(isinstance(m, torch.nn.BatchNorm1d) for m in model.output_layer.modules())
    _freeze_encoder(model)
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    model.eval()
    model.inference_descriptions_per_label = 2
    with torch.no_grad():
        ev, _ = model(sequence_onehots=x, sequence_lengths=lens,
                      label_embeddings=torch.from_numpy(g["label_embeddings"]).to(DEV))
    np.testing.assert_allclose(ev.cpu().numpy(), g["eval/logits_ens2"], atol=5e-4, rtol=1e-4)
    model.train()
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV)
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous().to(DEV)
    y = torch.from_numpy(g["multihots"]).to(DEV)
    u = torch.from_numpy(g["train/noise_u"]).to(DEV)
    replay_label_noise(monkeypatch, lambda t, *a, **k: u.clone())
    cfg = {"params": {"LOSS_FN": loss, "FOCAL_LOSS_GAMMA": 2, "FOCAL_LOSS_ALPHA": -1, "LABEL_SMOOTHING": 0.0}}
    loss_fn = get_loss(cfg, bce_pos_weight=torch.tensor(1.0))
    opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
    logits, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab, label_token_counts=cnt)
    l = loss_fn(logits, y.float())
    l.backward()
    p = f"train_{loss}/"
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[p + "logits"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(l.item(), float(g[p + "loss"]), rtol=1e-4)
    named = dict(model.named_parameters())
    seen_bias = 0
    for k in g.files:
        if k.startswith(p + "grad/"):
            name = k[len(p + "grad/"):]
            ref = g[k]
            np.testing.assert_allclose(named[name].grad.cpu().numpy(), ref, atol=2e-5 + 2e-4 * np.abs(ref).max(),
                                       err_msg=name)
            seen_bias += name.startswith("output_layer") and name.endswith(".bias")
    assert seen_bias == 4
    opt.step()
    np.testing.assert_allclose(opt.last_grad_norm.item(), float(g[p + "grad_norm"]), rtol=2e-4)
    got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k in g.files:
        if k.startswith(p + "sd_after/output_layer"):
            _assert_adam_close(got[k[len(p + "sd_after/"):]], g[k], k)


def test_embedding_dropouts(golden_dir):
    """SEQUENCE_EMBEDDING_DROPOUT / LABEL_EMBEDDING_DROPOUT (reference ProtNote.py:83-86 wraps W_p / W_l in
    Sequential(Dropout, MLP): checkpoint keys move to W_p.1.*): same logits and gradients as the dropout-free model
    fed the masked, rescaled rows (masks regenerated from the same device RNG state); eval ignores the dropouts."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(3)
    kw = dict(protein_embedding_dim=32, label_embedding_dim=16, latent_dim=16, output_mlp_hidden_dim_scale_factor=2,
              output_mlp_num_layers=2, projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=2)
    plain = ProtNote(**kw).to(DEV)
    drop = ProtNote(sequence_embedding_dropout=0.25, label_embedding_dropout=0.5, **kw).to(DEV)
    assert "W_p.1.0.weight" in drop.state_dict() and "W_l.1.0.weight" in drop.state_dict()
    drop.load_state_dict({k.replace("W_p.", "W_p.1.").replace("W_l.", "W_l.1."): v for k, v in plain.state_dict().items()})
    B, NL = 9, 21
    P_f = torch.randn(B, 32, generator=gen).to(DEV)
    lab = torch.randn(NL, 16, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.3).float().to(DEV)
    plain.train()
    drop.train()
    torch.manual_seed(11)
    out_d, _ = drop(sequence_embeddings=P_f, label_embeddings=lab)
    BCEWithLogitsLoss()(out_d, y).backward()
    torch.manual_seed(11)   # the two masks, in the order forward_train draws them
    P_m = torch.nn.functional.dropout(P_f, 0.25, training=True)
    L_m = torch.nn.functional.dropout(lab, 0.5, training=True)
    assert (P_m == 0).any() and (L_m == 0).any()
    out_p, _ = plain(sequence_embeddings=P_m, label_embeddings=L_m)
    BCEWithLogitsLoss()(out_p, y).backward()
    np.testing.assert_allclose(out_d.detach().cpu().numpy(), out_p.detach().cpu().numpy(), atol=1e-6)
    gp = dict(plain.named_parameters())
    for n, p in drop.named_parameters():
        ref = gp[n.replace("W_p.1.", "W_p.").replace("W_l.1.", "W_l.")].grad
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.cpu().numpy(), atol=1e-7 + 1e-5 * ref.abs().max().item())
    plain.eval()
    drop.eval()
    with torch.no_grad():
        a, _ = drop(sequence_embeddings=P_f, label_embeddings=lab)
        b, _ = plain(sequence_embeddings=P_f, label_embeddings=lab)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-6)


def test_extra_losses_golden(golden_dir):
    """LOSS_FN = RGDBCE / BatchWeightedBCE / WeightedBCE / CBLoss through get_loss: value and d/dlogits against the
    reference, with the TP/FN/FP counting still fused into the same pass."""
    from protnote_amd.utils.losses import get_loss

    g = _g(golden_dir, "losses_extra.npz")
    logits = torch.from_numpy(g["logits"]).to(DEV)
    y = torch.from_numpy(g["multihots"]).to(DEV)
    lw, lc = torch.from_numpy(g["label_weights"]), torch.from_numpy(g["label_counts"])
    cases = {"RGDBCE": ({"LOSS_FN": "RGDBCE", "RGDBCE_TEMP": 0.12}, None),
             "RGDBCE_hot": ({"LOSS_FN": "RGDBCE", "RGDBCE_TEMP": 5.0}, None),
             "BatchWeightedBCE": ({"LOSS_FN": "BatchWeightedBCE"}, None),
             "WeightedBCE": ({"LOSS_FN": "WeightedBCE"}, lw), "CBLoss": ({"LOSS_FN": "CBLoss"}, lc)}
    for name, (params, w) in cases.items():
        for tgt in (y.float(), y):               # float and int64 targets
            fn = get_loss({"params": params}, label_weights=w)
            counts = torch.zeros(3, y.shape[1], device=DEV)
            fn.metric_counts, fn.decision_threshold = counts, 0.5
            lg = logits.clone().requires_grad_(True)
            l = fn(lg, tgt)
            l.backward()
            np.testing.assert_allclose(l.item(), float(g[name + "/loss"]), rtol=2e-6, err_msg=name)
            ref = g[name + "/dlogits"]
            np.testing.assert_allclose(lg.grad.cpu().numpy(), ref, atol=1e-9 + 2e-6 * np.abs(ref).max(), rtol=2e-5,
                                       err_msg=name)
            pred = (torch.sigmoid(logits) >= 0.5).float()
            np.testing.assert_array_equal(counts[0].cpu().numpy(), (pred * y).sum(0).cpu().numpy())
    # LOSS_FN: SupCon (losses.py:7-56): row-softmax loss; the protein without positives gives NaN gradients, as there
    for tgt in (y.float(), y):
        fn = get_loss({"params": {"LOSS_FN": "SupCon", "SUPCON_TEMP": 0.07}})
        lg = logits.clone().requires_grad_(True)
        l = fn(lg, tgt)
        l.backward()
        np.testing.assert_allclose(l.item(), float(g["SupCon/loss"]), rtol=2e-6)
        ref = g["SupCon/dlogits"]
        np.testing.assert_allclose(lg.grad.cpu().numpy(), ref, atol=1e-9 + 2e-6 * np.nanmax(np.abs(ref)), rtol=2e-5,
                                   equal_nan=True)


def _small_batch(g):
    return {"sequence_onehots": torch.from_numpy(g["x"]).to(DEV), "sequence_lengths": torch.from_numpy(g["lens"]).to(DEV),
            "label_embeddings": torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV),
            "label_multihots": torch.from_numpy(g["multihots"]).float().to(DEV)}


def test_trainer_steps_on_last_batch_of_epoch(golden_dir):
    """GRADIENT_ACCUMULATION_STEPS = 2 over an epoch of 3 batches: the reference steps when its global counter is a
    multiple of GA OR on the last batch of the loader (ProtNoteTrainer.py:741-743), so nothing accumulated leaks into
    the next epoch.  Two optimiser steps, an empty gradient buffer afterwards, the model back in train mode after
    evaluate() (:671), and a loss that is the mean over the batches."""
    from protnote_amd.models.ProtNoteTrainer import Trainer
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam

    g = _g(golden_dir, "protnote_small_concatenation.npz")
    m, _ = make_protnote(g, DEV)
    _freeze_encoder(m)
    opt = FusedClipAdam(head_parameters(m), lr=3e-4, max_norm=1.0)
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    tr = Trainer(m, loss_fn, opt, gradient_accumulation_steps=2)
    batch = _small_batch(g)
    out = tr.train_one_epoch([batch, batch, batch])
    assert opt.step_count == 2 and float(opt.flat_g.abs().sum()) == 0.0 and tr.training_step == 3
    assert np.isfinite(out["loss"]) and 0 < out["loss"] < 1.0  # each batch's loss is already divided by GA = 2
    ev = tr.evaluate([batch], with_map=False)
    assert m.training and np.isfinite(ev["loss"])


def test_second_forward_keeps_pending_backward_when_memory_allows(golden_dir, monkeypatch):
    """Torch semantics for overlapping graphs: a differentiable forward that starts while an earlier forward's backward is
    pending gets an activation store of its own (held by its autograd context), so BOTH backwards run, in either order,
    and give the gradients of their own batch.  When the device has no room for a second store (two bench-size stores
    cannot exist; simulated here) the later forward takes the shared store and the earlier backward raises instead of
    returning gradients of the wrong batch."""
    from protnote_amd.models import train_path
    from protnote_amd.utils.losses import get_loss

    g = _g(golden_dir, "protnote_small_concatenation.npz")
    m, _ = make_protnote(g, DEV)
    _freeze_encoder(m)
    m.label_embedding_noising_alpha = 0.0  # (no random draw: the runs below are compared bit for bit)
    m.train()
    b = _small_batch(g)
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    kw1 = dict(sequence_onehots=b["sequence_onehots"], sequence_lengths=b["sequence_lengths"],
               label_embeddings=b["label_embeddings"])
    kw2 = dict(kw1, label_embeddings=b["label_embeddings"].flip(0).contiguous())  # a different batch
    y1, y2 = b["label_multihots"], b["label_multihots"].flip(1).contiguous()
    params = [p for p in m.parameters() if p.requires_grad]

    def grads_of(kw, y):  # one forward, one backward: the reference run
        bufs = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
        gs = torch.autograd.grad(loss_fn(m(**kw)[0], y), params)
        m.load_state_dict(bufs, strict=False)  # (train-mode BatchNorm moved its buffers; the comparison runs restart there)
        return gs

    ref1, ref2 = grads_of(kw1, y1), grads_of(kw2, y2)
    for order in ((0, 1), (1, 0)):
        losses = [loss_fn(m(**kw1)[0], y1), loss_fn(m(**kw2)[0], y2)]  # two graphs alive at once
        got = [None, None]
        for i in order:
            got[i] = torch.autograd.grad(losses[i], params)
        for a, r in zip(got[0] + got[1], ref1 + ref2):
            assert torch.equal(a, r)
    assert not train_path._backward_pending(m)

    monkeypatch.setattr(train_path, "_try_private_store", lambda *a, **k: None)  # "no room for a second store"
    l1 = loss_fn(m(**kw1)[0], y1)
    l2 = loss_fn(m(**kw2)[0], y2)
    with pytest.raises(RuntimeError, match="another differentiable forward"):
        l1.backward()
    l2.backward()  # the latest forward is intact
    m.eval()  # a forward that keeps nothing does not invalidate anything
    with torch.no_grad():
        m(**kw1)
    m.train()
    l3 = loss_fn(m(**kw1)[0], y1)
    l3.backward()


def test_optimizer_state_interchange_with_torch_adam(golden_dir):
    """`optimizer_state_dict` of a checkpoint (reference utils/models.py:304-321,366-367) moves both ways between
    FusedClipAdam and torch.optim.Adam over the same parameter list: after two steps on identical gradients the
    moments agree, a torch-Adam state loads into the fused optimiser (and continues identically), and the fused
    state loads into torch.optim.Adam."""
    from protnote_amd.utils.optim import FusedClipAdam

    torch.manual_seed(0)
    shapes = [(8, 12), (8,), (5, 8), (3,)]
    ps_f = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    ps_t = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps_f]
    fused = FusedClipAdam(ps_f, lr=1e-2, max_norm=None)
    ref = torch.optim.Adam(ps_t, lr=1e-2)
    assert fused.state_dict()["state"] == {} and fused.state_dict()["param_groups"][0]["params"] == [0, 1, 2, 3]

    def step_both(seed):
        gen = torch.Generator().manual_seed(seed)
        for pf, pt in zip(ps_f, ps_t):
            gr = torch.randn(pt.shape, generator=gen)
            pt.grad = gr.clone()
            pf.grad.copy_(gr.to(DEV))
        fused.step()
        ref.step()

    step_both(1)
    step_both(2)
    sd_f, sd_t = fused.state_dict(), ref.state_dict()
    assert set(sd_f["state"]) == set(sd_t["state"]) and sd_f["param_groups"][0]["betas"] == (0.9, 0.999)
    for i in sd_t["state"]:
        assert float(sd_f["state"][i]["step"]) == float(sd_t["state"][i]["step"]) == 2.0
        np.testing.assert_allclose(sd_f["state"][i]["exp_avg"].cpu().numpy(), sd_t["state"][i]["exp_avg"].numpy(),
                                   rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(sd_f["state"][i]["exp_avg_sq"].cpu().numpy(), sd_t["state"][i]["exp_avg_sq"].numpy(),
                                   rtol=1e-4, atol=1e-8)
    # torch -> fused: a fresh fused optimiser resumed from torch's state continues like torch does
    ps_r = [torch.nn.Parameter(p.detach().to(DEV).clone()) for p in ps_t]
    resumed = FusedClipAdam(ps_r, lr=123.0, max_norm=None)
    resumed.load_state_dict(sd_t)
    assert resumed.step_count == 2 and resumed.lr == 1e-2
    ps_f[:] = ps_r
    fused = resumed
    step_both(3)
    for pr, pt in zip(ps_r, ps_t):
        np.testing.assert_allclose(pr.detach().cpu().numpy(), pt.detach().numpy(), rtol=1e-5, atol=1e-6)
    # fused -> torch
    ref2 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps_t], lr=1e-2)
    ref2.load_state_dict(fused.state_dict())
    assert float(ref2.state_dict()["state"][0]["step"]) == 3.0
    # a parameter re-assigned behind the optimiser's back is detected
    ps_r[0].data = ps_r[0].data.clone()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        resumed.step()
    resumed.repack()
    resumed.step()


def test_build_training_optimizer_ids_follow_the_reference_order(golden_dir):
    """The reference builds torch.optim.Adam from model.named_parameters() filtered by requires_grad
    (ProtNoteTrainer.py:199-231): sequence_encoder (classifier included) before W_p, W_l, raw_attn_scorer, output_layer.
    build_training gives the fused optimiser's state the same ids, so with LABEL_EMBEDDING_POOLING_METHOD 'all' and
    TRAIN_SEQUENCE_ENCODER a reference `optimizer_state_dict` loads into it by id (and the other way round) although the
    fused optimiser's own flat layout orders the tensors differently and does not own the never-trained classifier.
    Also: the Trainer passes tokenized_labels through, so the pooling-'all' configuration trains end to end."""
    from protnote_amd.utils.configs import build_training

    g = _g(golden_dir, "protnote_small_attention.npz")
    model, _ = make_protnote(g, DEV, train_sequence_encoder=True)
    cfg = {"params": {"LOSS_FN": "BCE", "BCE_POS_WEIGHT": 1, "OPTIMIZER": "Adam", "LEARNING_RATE": 1e-3, "CLIP_VALUE": 1,
                      "GRADIENT_ACCUMULATION_STEPS": 1, "DECISION_TH": 0.5, "TRAIN_SEQUENCE_ENCODER": True}}
    loss_fn, opt, trainer = build_training(cfg, model)
    ref_list = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    names = [n for n, _ in ref_list]
    assert names[0].startswith("sequence_encoder.") and "sequence_encoder.output_layer.weight" in names
    assert names.index("raw_attn_scorer.weight") < names.index("output_layer.0.weight")
    pos = {id(p): k for k, (_, p) in enumerate(ref_list)}
    assert opt.n_param_ids == len(ref_list) and [pos[id(p)] for p in opt.params] == opt.param_ids
    batch = {"sequence_onehots": torch.from_numpy(g["x"]).to(DEV), "sequence_lengths": torch.from_numpy(g["lens"]).to(DEV),
             "label_embeddings": torch.from_numpy(g["hidden"]).to(DEV),
             "tokenized_labels": {"attention_mask": torch.from_numpy(g["attention_mask"]).to(DEV)},
             "label_multihots": torch.from_numpy(g["multihots"]).float().to(DEV)}
    out = trainer.train_one_epoch([batch])  # pooling 'all' through the Trainer (needs the attention mask)
    assert np.isfinite(out["loss"]) and opt.step_count == 1
    ev = trainer.evaluate([batch], estimate_map=True)
    assert 0.0 <= ev["map_micro"] <= 1.0
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == list(range(len(ref_list)))
    cls = [k for k, (n, _) in enumerate(ref_list) if n.startswith("sequence_encoder.output_layer.")]
    assert cls and all(k not in sd["state"] for k in cls)  # torch keeps no state for parameters that never get a gradient
    for k, (n, p) in enumerate(ref_list):
        if k not in cls:
            assert tuple(sd["state"][k]["exp_avg"].shape) == tuple(p.shape), n
    # the reference's own optimiser over the same list accepts it, and its state loads back by id
    tref = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for _, p in ref_list], lr=1e-3)
    tref.load_state_dict({"state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                          "param_groups": sd["param_groups"]})
    back = tref.state_dict()
    m0 = opt.flat_m.clone()
    opt.flat_m.zero_()
    opt.load_state_dict(back)
    assert torch.equal(opt.flat_m, m0) and opt.step_count == 1


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("fusion", ["concatenation", "concatenation_prod", "similarity"])
def test_train_step_bit_reproducible(golden_dir, fusion, math_mode):
    """Two optimisation steps from identical state give bit-identical loss, gradient norm, parameters, Adam moments and
    BN buffers, in both arithmetic modes: every cross-workgroup sum is a fixed-order reduction of per-workgroup
    partials (train-mode BN statistics, BN-backward S1/S2, loss mean, gradient norm, split-K weight gradients)."""
    import protnote_amd
    from protnote_amd.models.ProtNoteTrainer import train_step
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam

    g = _g(golden_dir, f"protnote_small_{fusion}.npz")
    batch = _small_batch(g)
    loss_fn = get_loss({"params": {"LOSS_FN": "FocalLoss", "FOCAL_LOSS_GAMMA": 2, "FOCAL_LOSS_ALPHA": -1,
                                   "LABEL_SMOOTHING": 0.0}}, bce_pos_weight=torch.tensor(1.0))

    def run():
        m, _ = make_protnote(g, DEV)
        m.label_embedding_noising_alpha = 0.0
        for n, p in m.named_parameters():
            p.requires_grad = True  # trainable encoder too: its backward reductions are covered as well
        m.train_sequence_encoder = True
        m.train()
        opt = FusedClipAdam(head_parameters(m) + list(m.sequence_encoder.trunk_parameters()), lr=3e-4, max_norm=1.0)
        out = []
        for _ in range(2):
            loss = train_step(m, loss_fn, opt, batch)
            out += [loss.clone(), opt.last_grad_norm.clone()]
        bufs = torch.cat([b.detach().float().reshape(-1) for b in m.buffers()])
        return out + [opt.flat_w.clone(), opt.flat_m.clone(), opt.flat_v.clone(), bufs]

    protnote_amd.set_math_mode(math_mode)
    try:
        a, b = run(), run()
    finally:
        protnote_amd.set_math_mode("f32")
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), (i, (x - y).abs().max().item())


def test_attention_pooling_golden(golden_dir, monkeypatch):
    """LABEL_EMBEDDING_POOLING_METHOD: all against the reference-generated vectors (ProtNote.py:89-91,154-166,266-267):
    inference pooling + logits, and one train step in which raw_attn_scorer is trained - its gradient comes from
    pn_additive_attention_bwd fed by the W_l input gradient; label noise scaled by alpha / sqrt(T) as in the reference."""
    from protnote_amd.models.train_path import trainable_parameters
    from protnote_amd.utils.losses import get_loss
    from protnote_amd.utils.optim import FusedClipAdam

    g = _g(golden_dir, "protnote_small_attention.npz")
    model, _ = make_protnote(g, DEV)
    _freeze_encoder(model)
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    hidden, mask = torch.from_numpy(g["hidden"]).to(DEV), torch.from_numpy(g["attention_mask"]).to(DEV)
    tok = {"attention_mask": mask}
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model.additive_attention(hidden, mask).cpu().numpy(), g["eval/pooled"], atol=2e-5,
                                   rtol=1e-4)
        lg, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=hidden, tokenized_labels=tok)
    np.testing.assert_allclose(lg.cpu().numpy(), g["eval/logits"], atol=5e-4, rtol=1e-4)

    model.train()
    u = torch.from_numpy(g["train/noise_u"]).to(DEV)
    replay_label_noise(monkeypatch, lambda t, *a, **k: u.clone())
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    params = trainable_parameters(model)
    assert any(p is model.raw_attn_scorer.weight for p in params)
    opt = FusedClipAdam(params, lr=3e-4, max_norm=1.0)
    logits, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=hidden, tokenized_labels=tok,
                      label_token_counts=torch.from_numpy(g["token_counts"]).to(DEV))
    l = loss_fn(logits, torch.from_numpy(g["multihots"]).to(DEV).float())
    l.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["train_BCE/logits"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(l.item(), float(g["train_BCE/loss"]), rtol=1e-4)
    named = dict(model.named_parameters())
    checked = 0
    for k in g.files:
        if k.startswith("train_BCE/grad/"):
            name = k[len("train_BCE/grad/"):]
            ref = g[k]
            np.testing.assert_allclose(named[name].grad.cpu().numpy(), ref, atol=2e-5 + 2e-4 * np.abs(ref).max(),
                                       err_msg=name)
            checked += name.startswith("raw_attn_scorer")
    assert checked == 2 and np.abs(g["train_BCE/grad/raw_attn_scorer.weight"]).max() > 1e-4
    opt.step()
    np.testing.assert_allclose(opt.last_grad_norm.item(), float(g["train_BCE/grad_norm"]), rtol=2e-4)
    got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for name in ("raw_attn_scorer.weight", "W_l.0.weight", "output_layer.11.weight"):
        _assert_adam_close(got[name], g["train_BCE/sd_after/" + name], name)
    # the scorer's bias has an exactly-zero true gradient (softmax is shift invariant; the reference's autograd gives
    # -4.7e-9): Adam turns that rounding noise into a +-lr move whose sign is arbitrary - only the magnitude is checked
    assert abs(got["raw_attn_scorer.bias"] - g["sd/raw_attn_scorer.bias"]).max() <= 3e-4 * 1.01


# 66 560 rows: even slab count per split; 66 624: odd + ragged last split; B = 8 / 20: batch sizes that are not multiples
# of 32 - the specialised weight-gradient kernel decodes (protein, label) per staged row on the scalar unit (ANYB); the
# reference ships per-GPU batches of 8 and 32 (configs/base_config.yaml:5-9)
@pytest.mark.parametrize("B,NL", [(64, 1040), (96, 694), (8, 8320), (20, 3400)])
def test_lds_dma_engine_bit_identical_train_step(B, NL):
    """Full-width head on a 64 x 1040 pair grid (66 560 rows: the smallest grid the LDS-DMA kernels take; B % 32 == 0 but
    not 256, so a 256-row tile spans several labels and the specialised TN kernel's scalar pair decode wraps) - logits and
    every gradient of a train step with the LDS-DMA GEMMs equal, bit for bit, those of the register-staged engine that
    the small-grid oracle tests pin (same products, same accumulation order; only the operand staging differs)."""
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(77)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.05).float().to(DEV)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).train()

    def run(dma):
        L.check(L.lib().pn_set_f32_dma(dma))
        for p in model.parameters():
            p.grad = None
        logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        BCEWithLogitsLoss()(logits, y).backward()
        return [logits.detach().clone()] + [p.grad.clone() for p in model.parameters()]

    try:
        a, b = run(1), run(0)
    finally:
        L.lib().pn_set_f32_dma(1)
    assert float(a[0].std()) > 0.1
    for i, (x, z) in enumerate(zip(a, b)):
        assert torch.equal(x, z), (i, (x - z).abs().max().item())


@pytest.mark.parametrize("R,M,N", [(65536, 256, 256), (131072 + 64, 512, 256), (70016, 256, 768)])
def test_gemm_tn_lds_dma_vs_f64_and_register_engine(R, M, N):
    """Weight-gradient contraction with the A operand staged by LDS-DMA (gemm_tn.hpp, ADMA: big tiles, R % 32 == 0)
    against f64 and, bit for bit, against the register-staged kernel (same products in the same order)."""
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(R + M)
    A = torch.randn(R, M, generator=g).to(DEV)
    Bm = (torch.randn(R, N, generator=g) + torch.arange(N) * 0.01).to(DEV)
    ref = A.double().T @ Bm.double()
    ws = torch.empty(16 * M * N * 4 + 1024, dtype=torch.uint8, device=DEV)

    def run(dma):
        L.check(L.lib().pn_set_f32_dma(dma))
        out = torch.full((M, N), float("nan"), device=DEV)
        L.check(L.lib().pn_gemm_tn(L.ptr(A), M, L.ptr(Bm), N, L.ptr(out), N, R, M, N, L.ptr(ws), ws.numel(),
                                   L.stream_ptr()))
        torch.cuda.synchronize()
        return out

    try:
        got, old = run(1), run(0)
    finally:
        L.lib().pn_set_f32_dma(1)
    assert (got.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item() * R ** 0.5 / 8
    assert torch.equal(got, old) and torch.equal(run(1), got)


def _dropout_masks(L, seed, p, B, NL, mlp_dims, pair_h, n_pair):
    """The masks the kernels generate (pn_dropout_mask), scaled by 1 / (1 - p), in the oracle's key / row convention."""
    def mask(stream, rows, cols):
        out = torch.empty(rows, cols, dtype=torch.float32, device=DEV)
        L.check(L.lib().pn_dropout_mask(seed, stream, p, rows, cols, L.ptr(out), L.stream_ptr()))
        return out.cpu() / (1.0 - p)

    masks = {}
    for prefix, base, rows in (("W_p.", 100, B), ("W_l.", 200, NL)):
        for n, h in enumerate(mlp_dims[:-1]):
            masks[f"{prefix}{n}"] = mask(base + n, rows, h)
        masks[f"{prefix}out"] = mask(base + 99, rows, mlp_dims[-1])
    for n in range(n_pair - 1):  # label-major pair rows r = j * B + i -> the joint tensor's protein-major i * NL + j
        m = mask(300 + n, NL * B, pair_h)
        masks[f"output_layer.{n}"] = m.view(NL, B, pair_h).permute(1, 0, 2).reshape(B * NL, pair_h).contiguous()
    return masks


@pytest.mark.parametrize("fusion", ["concatenation", "concatenation_prod"])
def test_mlp_dropout_train_step_vs_oracle(fusion):
    """OUTPUT_MLP_DROPOUT > 0 in training (reference ProtNote.py:63-81 via torchvision MLP, :369-371 get_mlp): Dropout
    after every hidden ReLU of W_p / W_l AND after their last Linear, and after every hidden layer of the output MLP but
    the last.  The kernels draw their masks from a counter-based hash; pn_dropout_mask hands the SAME masks to the CPU
    oracle, so logits and every gradient can be compared exactly (the RNG stream itself cannot match torch's)."""
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    p = 0.3
    kw = dict(protein_embedding_dim=32, label_embedding_dim=16, latent_dim=16, output_mlp_hidden_dim_scale_factor=2,
              output_mlp_num_layers=3, projection_head_num_layers=3, projection_head_hidden_dim_scale_factor=2,
              feature_fusion=fusion)
    torch.manual_seed(2)
    model = ProtNote(dropout=p, **kw).to(DEV)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.3)
    B, NL = 9, 21
    P_f = torch.randn(B, 32, generator=g)
    lab = torch.randn(NL, 16, generator=g)
    y = (torch.rand(B, NL, generator=g) < 0.3).float()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    model.train()
    torch.manual_seed(5)
    logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    BCEWithLogitsLoss()(logits, y.to(DEV)).backward()
    torch.manual_seed(5)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())  # the draw _HeadsTrainFn.forward made

    masks = _dropout_masks(L, seed, p, B, NL, [32, 32, 16], 32, 3)
    keep = torch.cat([m.flatten() for m in masks.values()])
    assert abs(float((keep > 0).float().mean()) - (1 - p)) < 0.02            # Bernoulli(1 - p)
    assert not torch.equal(masks["W_l.0"][:B], masks["W_p.0"])                # independent streams
    names = O.trainable_names(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    ref = O.protnote_forward(work, None, None, lab, fusion=fusion, training=True, sequence_embeddings=P_f,
                             dropout_masks=masks)
    ref_loss = O.bce_loss(ref, y)
    ref_grads = dict(zip(names, torch.autograd.grad(ref_loss, [leaves[k] for k in names])))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5, rtol=1e-4)
    for name, q in model.named_parameters():
        r = ref_grads[name].numpy()
        np.testing.assert_allclose(q.grad.cpu().numpy(), r, atol=1e-6 + 2e-4 * np.abs(r).max(), err_msg=name)

    # a different seed gives different masks; eval ignores dropout altogether
    torch.manual_seed(6)
    logits2, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    assert (logits2 - logits).abs().max().item() > 1e-3
    model.eval()
    plain = ProtNote(dropout=0.0, **kw).to(DEV).eval()
    plain.load_state_dict(model.state_dict())
    with torch.no_grad():
        a, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
        b, _ = plain(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    assert torch.equal(a, b)


def test_mlp_dropout_big_kernels_match_small_tiles():
    """Full-width head, 64 x 1040 pair grid with dropout 0.1: the 256-tile LDS-DMA kernels with the mask in their
    operand loaders (NT forward, TN weight gradient) against the 128-tile register-staged kernels on the same seed -
    two tilings of the same masked arithmetic."""
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(79)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 64, 1040
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.05).float().to(DEV)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, dropout=0.1)
    model.load_state_dict(sd)
    model = model.to(DEV).train()

    def run(dma):
        L.check(L.lib().pn_set_f32_dma(dma))
        for q in model.parameters():
            q.grad = None
        torch.manual_seed(123)
        logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        BCEWithLogitsLoss()(logits, y).backward()
        return [logits.detach().clone()] + [q.grad.clone() for q in model.parameters()]

    try:
        a, b = run(1), run(0)
    finally:
        L.lib().pn_set_f32_dma(1)
    model.mlp_dropout = 0.0
    c = run(1)
    assert (c[0] - a[0]).abs().max().item() > 1e-2  # dropout really changes the logits
    for i, (x, z) in enumerate(zip(a, b)):  # different tilings / split-K groupings: f32 reassociation noise only
        rel = (x - z).norm().item() / max(z.norm().item(), 1e-30)
        assert rel < 2e-4, (i, rel)


@pytest.mark.parametrize("NL", [20000, 20006])  # whole 32-row slabs (LDS-DMA / specialised TN kernels) and a ragged row count
def test_row_mlp_backward_big_kernels_match_small_tiles(NL):
    """W_l over a label table of >= 16384 rows: the backward materialises dY once and runs the 256-tile LDS-DMA NT kernel and
    the big TN tiles instead of regenerating dY in the operand loaders of the 128-tile engine.  With pn_set_mlp_materialize(0) the
    old backward runs (the one the small-size oracle tests pin) behind the same forward: the materialised dY equals the
    regenerated one and the NT products keep their order, so only the weight gradients' row splits differ -
    accumulation-order noise."""
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(83)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B = 4
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.05).float().to(DEV)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, label_embedding_noising_alpha=0.0)
    model.load_state_dict(sd)
    model = model.to(DEV).train()

    def run(mat):
        L.check(L.lib().pn_set_mlp_materialize(mat))
        for p in model.parameters():
            p.grad = None
        logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        BCEWithLogitsLoss()(logits, y).backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        (la, ga), (lb, gb) = run(1), run(0)
    finally:
        L.lib().pn_set_mlp_materialize(1)
    assert torch.equal(la, lb)  # same forward
    checked = 0
    for n in ga:
        if n.startswith("W_l."):
            rel = ((ga[n] - gb[n]).double().norm() / gb[n].double().norm().clamp_min(1e-30)).item()
            assert rel < 2e-5, (n, rel)
            checked += 1
    assert checked >= 7


OPT_CASES = ("adam", "adam_frozen_head", "adamw", "sgd", "sgd_frozen_head")


def _run_optimizer_case(g, case, monkeypatch):
    import json

    from protnote_amd.utils.configs import build_training

    params = json.loads(str(g[case + "/params_json"]))
    model, _ = make_protnote(g, DEV)
    loss_fn, opt, trainer = build_training({"params": params}, model)
    lab = torch.from_numpy(g["label_embeddings"]).to(DEV)
    cnt = torch.from_numpy(g["label_token_counts"]).to(DEV)
    n = int(g["n_batches"])
    loader = [{"sequence_onehots": torch.from_numpy(g[f"batch{k}/x"]).to(DEV),
               "sequence_lengths": torch.from_numpy(g[f"batch{k}/lens"]).to(DEV),
               "label_multihots": torch.from_numpy(g[f"batch{k}/multihots"]).to(DEV),
               "label_embeddings": lab, "label_token_counts": cnt} for k in range(n)]
    queue = [torch.from_numpy(g[f"batch{k}/noise_u"]).to(DEV) for k in range(n)]
    replay_label_noise(monkeypatch, lambda t, *a, **k: queue.pop(0))
    seen = []
    h = model.register_forward_hook(lambda m, a, out: seen.append(out[0].detach().clone()))
    metrics = trainer.train_one_epoch(loader)
    h.remove()
    return params, model, opt, metrics, seen


@pytest.mark.parametrize("case", OPT_CASES)
def test_optimizer_branches_golden(golden_dir, case, monkeypatch):
    """ProtNoteTrainer._set_optimizer's branches (ProtNoteTrainer.py:199-245) through build_training, against an epoch
    that the REFERENCE'S OWN ProtNoteTrainer.train_one_epoch ran on CPU (tests/golden/make_golden.py::
    golden_optimizer_branches): the set of parameters left trainable by TRAIN_PROJECTION_HEAD: False (output_layer.*
    frozen; W_p / W_l keep training - the reference's startswith quirk), OPTIMIZER Adam / AdamW / SGD with WEIGHT_DECAY,
    every batch's loss, the epoch metrics and the state dict after 10 optimisation steps."""
    g = _g(golden_dir, "optimizer_branches.npz")
    params, model, opt, metrics, seen = _run_optimizer_case(g, case, monkeypatch)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == [str(v) for v in g[case + "/trainable_names"]]
    assert opt.n_param_ids == int(g[case + "/n_optimizer_params"]) == len(opt.params)
    assert type(opt).__name__ == {"Adam": "FusedClipAdam", "AdamW": "FusedClipAdam", "SGD": "FusedClipSGD"}[params["OPTIMIZER"]]
    assert str(g[case + "/optimizer_class"]) == params["OPTIMIZER"]
    np.testing.assert_allclose(seen[0].cpu().numpy(), g[case + "/first_logits"], atol=5e-4, rtol=1e-4)
    sgd = params["OPTIMIZER"] == "SGD"
    np.testing.assert_allclose(metrics["loss"], float(g[case + "/metrics/train_loss"]), rtol=1e-4 if sgd else 3e-3)
    if sgd:  # (Adam's sign-like update lets a handful of threshold decisions differ)
        np.testing.assert_allclose(metrics["f1_macro"], float(g[case + "/metrics/train_f1_macro"]), rtol=1e-5)
        np.testing.assert_allclose(metrics["f1_micro"], float(g[case + "/metrics/train_f1_micro"]), rtol=1e-5)
    lr, n = float(params["LEARNING_RATE"]), int(g["n_batches"])
    got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    head = bool(params["TRAIN_PROJECTION_HEAD"])
    for k in g.files:
        if not k.startswith(case + "/sd_after/"):
            continue
        name = k[len(case + "/sd_after/"):]
        if not head and name.startswith("output_layer") and not name.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert np.array_equal(got[name], g["sd/" + name]), name  # frozen: bit-for-bit the initial weights
            assert dict(model.named_parameters())[name].grad is None, name
        elif sgd or name.endswith(("running_mean", "running_var", "num_batches_tracked")) or name.startswith("sequence_encoder"):
            np.testing.assert_allclose(got[name], g[k], atol=5e-5, rtol=5e-4, err_msg=name)
        else:
            _assert_adam_close(got[name], g[k], name, lr=lr, steps=n, frac=0.98)


def test_frozen_output_layer_same_forward_and_remaining_gradients(golden_dir, monkeypatch):
    """Freezing output_layer changes what the backward COMPUTES (no dW / dgamma / dbeta / dw_out / db_out for it: NULL
    destinations in pn_pairhead_grads), never the numbers that remain: logits and the W_p / W_l gradients of a step with the
    head frozen are bit-identical to the unfrozen step's."""
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    g = _g(golden_dir, "protnote_small_concatenation.npz")
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV)
    y = torch.from_numpy(g["multihots"]).float().to(DEV)
    res = []
    for frozen in (False, True):
        model, _ = make_protnote(g, DEV)
        _freeze_encoder(model)
        model.label_embedding_noising_alpha = 0.0
        if frozen:
            for n_, p_ in model.named_parameters():
                if n_.startswith("output_layer"):
                    p_.requires_grad = False
        model.train()
        logits, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab)
        BCEWithLogitsLoss()(logits, y).backward()
        res.append((logits.detach().clone(), {n_: p_.grad for n_, p_ in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for n_, gr in res[1][1].items():
        if n_.startswith("output_layer") or n_.startswith("sequence_encoder"):
            assert gr is None, n_
        else:
            assert torch.equal(gr, res[0][1][n_]), n_


def test_sgd_state_interchange_with_torch_sgd(golden_dir):
    """FusedClipSGD.state_dict() has torch.optim.SGD's layout (a reference checkpoint's optimizer_state_dict under
    OPTIMIZER: SGD, utils/models.py:304-321) and one fused step equals torch's SGD step on the same gradients."""
    from protnote_amd.utils.optim import FusedClipSGD

    gen = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(s, generator=gen).to(DEV)) for s in ((7, 5), (13,), (4, 4, 3))]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = FusedClipSGD(ps, lr=0.05, weight_decay=0.01, max_norm=1.0)
    tref = torch.optim.SGD(ref, lr=0.05, weight_decay=0.01)
    for step in range(3):
        for p, r in zip(ps, ref):
            gr = torch.randn(p.shape, generator=gen)
            p.grad.copy_(gr.to(DEV))
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        tref.step()
        opt.step()
        for p, r in zip(ps, ref):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), atol=1e-6, rtol=1e-6)
    sd = opt.state_dict()
    # momentum 0: torch's SGD keeps NO per-parameter state, and neither does the fused one (round 4 emitted
    # {'momentum_buffer': None} per parameter: it loaded, but the layout was not torch's)
    assert sd["state"] == {} and tref.state_dict()["state"] == {}
    assert opt.flat_m is None and opt.flat_v is None
    tref.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})  # torch accepts the layout
    opt.load_state_dict(tref.state_dict())
    # with momentum the velocity block exists, travels both ways, and "first step or not" survives the round trip
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref2 = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt2 = FusedClipSGD(ps2, lr=0.05, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    tref2 = torch.optim.SGD(ref2, lr=0.05, momentum=0.9, weight_decay=0.01)
    for step in range(3):
        for p, r in zip(ps2, ref2):
            gr = torch.randn(p.shape, generator=gen)
            p.grad.copy_(gr.to(DEV))
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref2, 1.0)
        tref2.step()
        opt2.step()
        for p, r in zip(ps2, ref2):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), atol=1e-6, rtol=1e-6)
    sd2 = opt2.state_dict()
    assert sorted(sd2["state"]) == [0, 1, 2] and opt2.flat_v is None
    for i, r in enumerate(ref2):
        np.testing.assert_allclose(sd2["state"][i]["momentum_buffer"].cpu().numpy(),
                                   tref2.state_dict()["state"][i]["momentum_buffer"].numpy(), atol=1e-6, rtol=1e-6)
    tref2.load_state_dict({"state": {k: {"momentum_buffer": v["momentum_buffer"].cpu()} for k, v in sd2["state"].items()},
                           "param_groups": sd2["param_groups"]})
    opt2.load_state_dict(tref2.state_dict())
    assert opt2.step_count == 1


def test_two_models_on_two_streams_concurrently_equal_serial():
    """The C ABI promises "no hidden allocation, thread-safe per stream" (include/protnote_hip.h).  Two full-width models
    take one train step each (forward, BCE, backward, clip + Adam) at the same time - two host threads, two streams of one
    device, grids large enough for every big kernel incl. the paced weight-gradient kernel whose arrival counters used to
    be one static buffer per device - and end with exactly the bits of the same two steps run one after the other."""
    _two_models_two_streams(modes=(None, None))


def test_two_models_on_two_streams_in_different_math_modes():
    """The arithmetic travels with the call (pn_*.math_mode / pn_pairhead.backward_math, set from `model.math_mode` /
    `model.backward_math`), not with the process: thread 0 trains a model in exact f32, thread 1 another one on the bf16x3
    forward + bf16 backward, AT THE SAME TIME, while a third thread keeps flipping the process defaults
    (protnote_amd.set_math_mode / set_backward_math) - and each ends with exactly the bits of its own serial run.  (Round 4's
    switches were process globals: a flip on one thread changed the other's next launch.)"""
    out = _two_models_two_streams(modes=(("f32", "same"), ("bf16x3", "bf16")), flip_defaults=True)
    # ... and the second model really ran another arithmetic than the first one's: its serial f32 run differs
    plain = _two_models_two_streams(modes=(("f32", "same"), ("f32", "same")), serial_only=True)
    assert torch.equal(out[0][0], plain[0][0])
    assert not torch.equal(out[1][0], plain[1][0]) and not torch.equal(out[1][1], plain[1][1])
    # (third-step logits after two Adam updates at lr 1e-3 on bf16-backward gradients: a different trajectory, same scale;
    #  measured 0.18 apart on logits of std ~1)
    assert (out[1][0] - plain[1][0]).abs().max().item() < 1.0


def _two_models_two_streams(modes, flip_defaults=False, serial_only=False):
    import threading

    import protnote_amd

    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import BCEWithLogitsLoss
    from protnote_amd.utils.optim import FusedClipAdam

    B, NL = 64, 2080  # 133 120 pair rows: 256-tile NT / specialised TN kernels, 2+ row splits
    cases = []
    for seed in (5, 6):
        gen = torch.Generator().manual_seed(seed)
        cases.append(dict(sd=random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3),
                          P_f=torch.randn(B, 1100, generator=gen).to(DEV), lab=torch.randn(NL, 1024, generator=gen).to(DEV),
                          y=(torch.rand(B, NL, generator=gen) < 0.05).float().to(DEV)))

    def build(c, mode):
        m = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
        m.load_state_dict(c["sd"])
        m = m.to(DEV).train()
        if mode is not None:
            m.math_mode, m.backward_math = mode
        return m, FusedClipAdam(head_parameters(m), lr=1e-3, max_norm=1.0)

    def step(m, opt, c, stream, out, k, reps=3):
        try:
            with torch.cuda.stream(stream):
                for _ in range(reps):
                    logits, _ = m(sequence_embeddings=c["P_f"], label_embeddings=c["lab"])
                    BCEWithLogitsLoss()(logits, c["y"]).backward()
                    opt.step()
                    opt.zero_grad()
                out[k] = (logits.detach().clone(), opt.flat_w.clone(), opt.flat_m.clone(),
                          [b.clone() for b in m.buffers()])
            stream.synchronize()
        except Exception as e:  # noqa: BLE001 - surfaced by the assert below
            out[k] = e

    serial, conc = {}, {}
    for k, c in enumerate(cases):
        m, opt = build(c, modes[k])
        step(m, opt, c, torch.cuda.current_stream(), serial, k)
    torch.cuda.synchronize()
    if serial_only:
        for k in range(2):
            assert not isinstance(serial[k], Exception), serial[k]
        return serial
    built = [build(c, modes[k]) for k, c in enumerate(cases)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    threads = [threading.Thread(target=step, args=(built[k][0], built[k][1], cases[k], streams[k], conc, k))
               for k in range(2)]
    stop = threading.Event()

    def flipper():  # the process defaults change under the two training threads
        k = 0
        while not stop.is_set():
            protnote_amd.set_math_mode(("bf16x3", "f32")[k & 1])
            protnote_amd.set_backward_math(("bf16", "same")[k & 1])
            k += 1
            time.sleep(0.001)

    fl = threading.Thread(target=flipper) if flip_defaults else None
    try:
        if fl is not None:
            fl.start()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        stop.set()
        if fl is not None:
            fl.join()
        protnote_amd.set_math_mode("f32")
        protnote_amd.set_backward_math("same")
    torch.cuda.synchronize()
    for k in range(2):
        assert not isinstance(conc[k], Exception), conc[k]
        assert not isinstance(serial[k], Exception), serial[k]
        a, b = serial[k], conc[k]
        assert float(a[0].std()) > 0.05
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), k
        for x, z in zip(a[3], b[3]):
            assert torch.equal(x, z), k
    return conc


def test_reference_ddp_checkpoint_resumes_on_the_fused_optimizer(golden_dir):
    """--from-checkpoint on the HIP path (bin/main.py:521-527 -> utils/models.py:324-374): a checkpoint the reference's
    save_checkpoint wrote from a DDP-wrapped model and its torch Adam is restored by the load_model twin into the twin
    model + FusedClipAdam built by build_training - weights, Adam moments by parameter id, step count, epoch, metric -
    and training continues from it."""
    import json

    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer
    from protnote_amd.utils import models as M
    from protnote_amd.utils.configs import build_training

    d = os.path.join(golden_dir, "disk_formats")
    c = json.load(open(os.path.join(d, "disk_formats.json")))["checkpoint"]
    exp = np.load(os.path.join(d, "expected.npz"))
    enc = ProteInfer(activation=torch.nn.ReLU, **c["enc_cfg"])
    model = ProtNote(sequence_encoder=enc, label_encoder=None, feature_fusion="concatenation", **c["head_cfg"]).to(DEV)
    cfg = {"params": {"LOSS_FN": "BCE", "BCE_POS_WEIGHT": 1, "OPTIMIZER": "Adam", "LEARNING_RATE": 1.0, "CLIP_VALUE": 1,
                      "TRAIN_SEQUENCE_ENCODER": False}}
    loss_fn, opt, trainer = build_training(cfg, model)
    M.load_model(trainer, os.path.join(d, c["file"]), rank=0, from_checkpoint=True)
    assert (trainer.epoch, trainer.starting_epoch, trainer.best_val_metric) == (7, 7, c["best_val_metric"])
    assert opt.step_count == 1 and opt.lr == c["optimizer_lr"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    for k, v in model.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), exp["ckpt/sd/" + k]), k
    off = dict((id(p), o) for p, o in opt._offsets())
    for i, (n, p) in enumerate(named):  # the reference's ids = position in the requires_grad-filtered list
        o = off[id(p)]
        assert np.array_equal(opt.flat_m[o:o + p.numel()].view_as(p).cpu().numpy(), exp[f"ckpt/opt/{i}/exp_avg"]), n
        assert p.data_ptr() == opt.flat_w.data_ptr() + 4 * o  # still views of the flat block after the load
    gen = torch.Generator().manual_seed(1)
    x = torch.nn.functional.one_hot(torch.randint(0, 20, (3, 12), generator=gen), 20).permute(0, 2, 1).float().to(DEV)
    batch = {"sequence_onehots": x, "sequence_lengths": torch.tensor([12, 7, 12]).to(DEV),
             "label_embeddings": torch.randn(6, 8, generator=gen).to(DEV),
             "label_multihots": (torch.rand(3, 6, generator=gen) < 0.4).float().to(DEV)}
    out = trainer.train_one_epoch([batch])
    assert np.isfinite(out["loss"]) and opt.step_count == 2
