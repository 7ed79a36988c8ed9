"""Opt-in bf16x3 arithmetic of the pair-grid GEMMs (pn_set_math_mode(1)): accuracy class against f64 and the 1e-3
logit bound of the north star; the default f32 mode is restored after every test."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import random_head_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def bf16x3():
    import protnote_amd

    protnote_amd.set_math_mode("bf16x3")
    yield
    protnote_amd.set_math_mode("f32")


def _gemm_nt(A, W, scale=None, shift=None):
    from protnote_amd import _lib

    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=DEV)
    _lib.check(_lib.lib().pn_gemm_nt(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(out), N, M, N, K, None,
                                     _lib.ptr(scale), _lib.ptr(shift), None, None, 0, None, 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out


def test_bf16x3_gemm_error_class(bf16x3):
    """C = relu(A*s+t) W^T on the split-bf16 path: error relative to sum |a||w| is ~1e-5 / sqrt(K)-ish, far below
    bf16 (4e-3) and above f32 (6e-8); ragged M exercises the row clamp."""
    import protnote_amd

    g = torch.Generator().manual_seed(0)
    M, N, K = 65536 + 37, 512, 160
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    s = (torch.rand(K, generator=g) + 0.5).to(DEV)
    t = (torch.randn(K, generator=g) * 0.1).to(DEV)
    got = _gemm_nt(A, W, s, t).double()
    act = torch.relu(A.double() * s.double() + t.double())
    ref = act @ W.double().T
    scale = (act.abs() @ W.double().abs().T)
    rel = ((got - ref).abs() / scale).max().item()
    assert rel < 2e-5, rel
    protnote_amd.set_math_mode("f32")
    exact = _gemm_nt(A, W, s, t).double()
    rel32 = ((exact - ref).abs() / scale).max().item()
    assert rel32 < 1e-6 and rel > 2 * rel32, (rel, rel32)   # i.e. the bf16x3 kernel really ran above


def test_bf16x3_logits_within_north_star(bf16x3):
    """Full-width eval forward (d=1024, h=3072, 3-layer head) on 96 x 700 pairs: logits within 1e-3 of the fp32 CPU
    oracle (north-star bound), measured error is ~1e-4."""
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(31)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 96, 700
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    with torch.no_grad():
        out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    err = (out.cpu() - ref).abs().max().item()
    assert ref.abs().max().item() > 0.5 and err < 1e-3, err
    print("bf16x3 max logit error", err)


def test_bf16x3_gemm_tn_error_class(bf16x3):
    """C = A^T B over a long, ragged row range (split-K, masked last slab) on the split-bf16 TN kernel."""
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(1)
    for R, M, N in ((65536 + 40, 256, 512), (131072 + 7, 512, 256)):
        A = torch.randn(R, M, generator=g)
        Bm = torch.randn(R, N, generator=g) + torch.arange(N) * 0.01
        Ad, Bd = A.to(DEV), Bm.to(DEV)
        ref = (Ad.double().T @ Bd.double()).cpu()
        scale = (Ad.double().abs().T @ Bd.double().abs()).cpu()
        Cd = torch.full((M, N), float("nan"), device=DEV)
        ws = torch.empty(16 * M * N * 4 + 1024, dtype=torch.uint8, device=DEV)
        L.check(L.lib().pn_gemm_tn(L.ptr(Ad), M, L.ptr(Bd), N, L.ptr(Cd), N, R, M, N, L.ptr(ws), ws.numel(),
                                   L.stream_ptr()))
        torch.cuda.synchronize()
        rel = ((Cd.cpu().double() - ref).abs() / scale).max().item()
        assert 1e-8 < rel < 2e-5, (R, M, N, rel)    # > f32 class: the bf16x3 kernel really ran


@pytest.mark.parametrize("B,NL,chunk", [(64, 1100, None), (72, 920, 300),
                                        (66, 1000, None)])  # B % 8 != 0: the pair-sum weight gradient stays on the f32 kernel
def test_bf16x3_train_step_vs_oracle(bf16x3, B, NL, chunk):
    """Full-width train step (d=1024, h=3072, 3 hidden layers) with every pair-grid GEMM (forward, dh, dW) on the
    split-bf16 path: logits within the 1e-3 north-star bound of the f64 oracle, loss to 1e-4, every gradient within
    1e-2 (Frobenius) - the f32 CPU path itself is 5e-4..2e-3 away at this grid size (ReLU-mask flips)."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(21)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()
    # float64 ground truth: the oracle's naive formulation in stock torch ops, evaluated on the device (same numbers to
    # ~1e-13 as on the host, seconds instead of half a minute)
    ref_sd = {k: (v.clone().double() if v.is_floating_point() else v.clone()).to(DEV) for k, v in sd.items()}
    names = O.trainable_names(ref_sd)
    leaves = {k: ref_sd[k].clone().requires_grad_(True) for k in names}
    work = dict(ref_sd)
    work.update(leaves)
    lg = O.protnote_forward(work, None, None, lab.double().to(DEV), training=True, sequence_embeddings=P_f.double().to(DEV))
    ls = O.bce_loss(lg, y.double().to(DEV))
    ref_grads = dict(zip(names, (g_.cpu() for g_ in torch.autograd.grad(ls, [leaves[k] for k in names]))))
    lg, ls = lg.detach().cpu(), ls.detach().cpu()
    del work, leaves, ref_sd
    torch.cuda.empty_cache()

    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    model.pair_label_chunk = chunk
    logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    l = BCEWithLogitsLoss()(logits, y.to(DEV))
    l.backward()
    err = (logits.detach().cpu().double() - lg.detach()).abs().max().item()
    assert err < 1e-3, err
    np.testing.assert_allclose(l.item(), ls.item(), rtol=1e-4)
    worst = 0.0
    for name, p in model.named_parameters():
        ref = ref_grads[name]
        rel = (p.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-30)
        worst = max(worst, rel)
        assert rel < 1e-2, (name, rel)
    print(f"bf16x3 train: logit err {err:.2e}, worst grad rel {worst:.2e}")


def test_bf16x3_training_tracks_f32():
    """Four optimisation steps (fwd + bwd + clip + Adam, lr 3e-4) of the full-width head on a 64 x 1100 pair grid, once
    in exact f32 and once with the bf16x3 GEMMs, from the same initial state: the loss trajectories and the logits of
    the final model stay together (same accuracy class as run-to-run f32 noise through Adam's sign-like updates)."""
    import protnote_amd
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import BCEWithLogitsLoss
    from protnote_amd.utils.optim import FusedClipAdam

    gen = torch.Generator().manual_seed(5)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 64, 1100
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float().to(DEV)

    def run(mode):
        protnote_amd.set_math_mode(mode)
        try:
            model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                             projection_head_hidden_dim_scale_factor=3)
            model.load_state_dict(sd)
            model = model.to(DEV).train()
            opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
            losses = []
            for _ in range(4):
                logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
                l = BCEWithLogitsLoss()(logits, y)
                l.backward()
                opt.step()
                opt.zero_grad()
                losses.append(l.item())
            model.eval()
            with torch.no_grad():
                final, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
            return np.array(losses), final.cpu().numpy()
        finally:
            protnote_amd.set_math_mode("f32")

    l32, f32 = run("f32")
    lb3, fb3 = run("bf16x3")
    assert l32[-1] < l32[0]                                   # it trains
    np.testing.assert_allclose(lb3, l32, rtol=2e-3)
    assert np.abs(fb3 - f32).max() < 2e-2 * max(1.0, np.abs(f32).max()), np.abs(fb3 - f32).max()
    print("losses f32", l32, "bf16x3", lb3, "final logit diff", np.abs(fb3 - f32).max())


def test_bf16x3_encoder_and_label_projection(bf16x3):
    """In bf16x3 mode the encoder's convolutions (implicit GEMM: tap gather, ragged K = 1100 / 550 / 20, ragged N) and
    the W_l row MLP over a large label set also run on the split-bf16 kernel (general path): embeddings vs the oracle
    (eval and train-mode BatchNorm statistics), label projection vs the oracle."""
    from tests.helpers import make_encoder, random_encoder_sd
    from protnote_amd.models.ProtNote import ProtNote

    cfg = dict(num_labels=11, input_channels=20, output_channels=1100, kernel_size=9, dilation_base=3,
               num_resnet_blocks=5, bottleneck_factor=0.5)
    gen = torch.Generator().manual_seed(11)
    sd = random_encoder_sd(cfg, gen)
    lens = [512, 1, 333, 512, 77, 500, 40, 511, 256, 129]
    ids = torch.randint(0, 20, (len(lens), 512), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    lens_t = torch.tensor(lens)
    ref = O.proteinfer_get_embeddings({k: v.clone() for k, v in sd.items()}, x, lens_t)
    enc = make_encoder(sd, "", cfg, DEV).eval()
    for p in enc.parameters():
        p.requires_grad = False
    emb = enc.get_embeddings(x.to(DEV), lens_t.to(DEV))
    err = (emb.cpu() - ref).abs().max().item()
    assert err < 5e-4 * max(1.0, ref.abs().max().item()), err
    # train-mode BatchNorm (batch statistics from the epilogue's column sums) + running-stat updates
    sd_t = {k: v.clone() for k, v in sd.items()}
    ref_t = O.proteinfer_get_embeddings(sd_t, x, lens_t, training=True)
    enc.train()
    emb_t = enc.get_embeddings(x.to(DEV), lens_t.to(DEV))
    err_t = (emb_t.cpu() - ref_t).abs().max().item()
    assert err_t < 5e-4 * max(1.0, ref_t.abs().max().item()), err_t
    got = {k: v.cpu() for k, v in enc.state_dict().items()}
    for k, v in sd_t.items():
        if k.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(got[k].numpy(), v.numpy(), atol=2e-4, rtol=2e-3, err_msg=k)
    # W_l over 9000 labels (row GEMMs with M >= 8192)
    hsd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(hsd)
    model = model.to(DEV).eval()
    lab = torch.randn(9000, 1024, generator=gen)
    with torch.no_grad():
        L_e = model._project_eval(model.W_l, lab.to(DEV)).cpu()
    ref_L = O.mlp_rows({k: v.clone() for k, v in hsd.items()}, "W_l.", lab, training=False)
    errL = (L_e - ref_L).abs().max().item()
    assert errL < 5e-4 * max(1.0, ref_L.abs().max().item()), errL
    print(f"bf16x3 encoder err {err:.2e} (train-BN {err_t:.2e}), W_l err {errL:.2e}")


def test_bf16x3_full_size_step_matches_f32():
    """BASELINE configs[2] size (B=256, L=512, N_L=32102, full width): the same train forward+backward in exact f32 and
    in bf16x3 mode - logits within the 1e-3 north-star bound of each other over all 8.2 M pairs, loss to 1e-5,
    gradients of the same accuracy class as two f32 runs of the same step (ReLU-mask flips over 2.5e10 activations)."""
    import protnote_amd
    from bench import build_model, synthetic_batch
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    dev = torch.device(DEV)
    model = build_model(dev, unit_scale_weights=True)
    model.label_embedding_noising_alpha = 0.0
    model.train()
    batch = synthetic_batch(256, 512, 32102, dev, seed=5)
    out = {}
    for mode in ("f32", "bf16x3"):
        protnote_amd.set_math_mode(mode)
        try:
            for p in model.parameters():
                p.grad = None
            logits, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                              label_embeddings=batch["label_embeddings"])
            loss = BCEWithLogitsLoss()(logits, batch["label_multihots"])
            loss.backward()
            out[mode] = (loss.item(), logits.detach().clone(),
                         {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            protnote_amd.set_math_mode("f32")
    (l0, lg0, g0), (l1, lg1, g1) = out["f32"], out["bf16x3"]
    assert lg0.abs().max().item() > 1.0
    err = (lg1 - lg0).abs().max().item()
    assert err < 1e-3, err
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0))
    worst = max((g1[n] - g0[n]).norm().item() / max(g0[n].norm().item(), 1e-30) for n in g0)
    assert worst < 2e-2, worst
    print(f"full size: max |logit diff| {err:.2e}, loss {l0:.6f} vs {l1:.6f}, worst grad rel {worst:.2e}")


@pytest.mark.parametrize("latent,scale", [(500, 2), (512, 2)])
def test_bf16x3_mode_other_widths(bf16x3, latent, scale):
    """Widths the split-bf16 kernels do not take (hidden = 1000: not a multiple of 256) silently stay on the exact f32
    kernels; widths they do take (hidden = 1024) run on them - either way the logits match the oracle."""
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(8)
    h = latent * scale
    sd = random_head_sd(gen, 1100, 1024, latent, h, 2, h, 2)
    B, NL = 64, 1100
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f)
    model = ProtNote(latent_dim=latent, output_mlp_hidden_dim_scale_factor=scale, output_mlp_num_layers=2,
                     projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=scale)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    with torch.no_grad():
        out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-3, err


def test_bf16x3_map_parity(bf16x3):
    """BASELINE metric 'mAP parity vs reference' in bf16x3 mode: micro / macro AP of the device metrics over the HIP
    logits (full-width head, 256 x 300 pairs: large enough for the split-bf16 kernels) against the AP of the CPU
    oracle's logits."""
    from oracle import metrics_oracle as MO
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    gen = torch.Generator().manual_seed(31)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 256, 300
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f)
    y = (torch.rand(B, NL, generator=gen) < torch.sigmoid(2 * ref - 2)).numpy()
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    with torch.no_grad():
        out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    assert (out.cpu() - ref).abs().max().item() < 1e-3
    acc = DeviceAveragePrecision(NL, B, DEV)
    acc.update(torch.sigmoid(out), torch.from_numpy(y).to(DEV))
    m = acc.compute()
    pr = torch.sigmoid(ref).numpy()
    mi_ref = MO.average_precision_fast(pr.ravel(), y.ravel())
    ma_ref = MO.macro_mean([MO.average_precision_fast(pr[:, j], y[:, j]) for j in range(NL)])
    assert 0.2 < mi_ref < 0.99
    assert abs(m["map_micro"] - mi_ref) < 1e-4 and abs(m["map_macro"] - ma_ref) < 2e-4, (m, mi_ref, ma_ref)


def test_bf16x3_weight_dma_bit_identical_train_step():
    """bf16x3 mode, 64 x 1040 pair grid at full width: staging the pre-split weight planes by LDS-DMA
    (pn_set_b3_dma(1), default) gives bit-identical logits and gradients to the register-staged kernels - the same
    hi / lo values meet the same products in the same order; only the path into the LDS differs."""
    import protnote_amd
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss
    from tests.helpers import random_head_sd

    gen = torch.Generator().manual_seed(78)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 64, 1040
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.05).float().to(DEV)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).train()

    def run(dma):
        L.check(L.lib().pn_set_b3_dma(dma))
        for p in model.parameters():
            p.grad = None
        logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        BCEWithLogitsLoss()(logits, y).backward()
        return [logits.detach().clone()] + [p.grad.clone() for p in model.parameters()]

    protnote_amd.set_math_mode("bf16x3")
    try:
        a, b = run(1), run(0)
        model.eval()
        with torch.no_grad():
            e1 = model(sequence_embeddings=P_f, label_embeddings=lab)[0]
            L.check(L.lib().pn_set_b3_dma(1))
            e2 = model(sequence_embeddings=P_f, label_embeddings=lab)[0]
    finally:
        L.lib().pn_set_b3_dma(1)
        protnote_amd.set_math_mode("f32")
    assert float(a[0].std()) > 0.1 and torch.equal(e1, e2)
    for i, (x, z) in enumerate(zip(a, b)):
        assert torch.equal(x, z), (i, (x - z).abs().max().item())
