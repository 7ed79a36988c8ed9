"""pn_set_backward_math(1): the hidden layers' backward pair-grid GEMMs (dW_l = dz_l^T h_{l-1}, dh_{l-1} = dz_l W_l) on ONE
bf16 product with f32 accumulation - the arithmetic class of the reference's own training run, whose Linear gradient
GEMMs execute in half precision under torch.autocast (ProtNoteTrainer.py:728-738).  The forward is untouched."""
import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import random_head_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def bwd_bf16():
    import protnote_amd

    protnote_amd.set_backward_math("bf16")
    yield
    protnote_amd.set_backward_math("same")


def _full_width_model(sd):
    from protnote_amd.models.ProtNote import ProtNote

    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    return model.to(DEV).train()


def _oracle_grads(sd, P_f, lab, y, dtype, autocast=False, fusion="concatenation"):
    """Loss, logits and gradients of the oracle's naive formulation (stock torch ops + autograd) on the device."""
    ref_sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()).to(DEV) for k, v in sd.items()}
    names = O.trainable_names(ref_sd)
    leaves = {k: ref_sd[k].clone().requires_grad_(True) for k in names}
    work = dict(ref_sd)
    work.update(leaves)
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else torch.autocast("cuda", enabled=False)
    with ctx:
        lg = O.protnote_forward(work, None, None, lab.to(dtype).to(DEV), training=True, fusion=fusion,
                                sequence_embeddings=P_f.to(dtype).to(DEV))
    ls = O.bce_loss(lg.to(dtype), y.to(dtype).to(DEV))
    grads = dict(zip(names, (g_.double().cpu() for g_ in torch.autograd.grad(ls, [leaves[k] for k in names]))))
    return lg.detach().double().cpu(), float(ls.item()), grads


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
def test_backward_bf16_step_vs_f64_with_autocast_yardstick(math_mode):
    """Full-width head, 256 x 300 pairs (76 800 rows: both the TN and the NT single-product kernels take their shapes).
    (i) logits and loss are BIT-identical to the same forward mode with the default backward; (ii) every gradient is
    within 2 x the error that the oracle's own formulation shows when torch runs it under autocast(bfloat16) on the
    device - the acceptance bar for an AMP-class backward - measured against the float64 oracle; (iii) the gradients
    differ from the default backward's (the single-product kernels really ran)."""
    import protnote_amd
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(33)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 256, 300
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()
    lg64, ls64, g64 = _oracle_grads(sd, P_f, lab, y, torch.float64)
    torch.cuda.empty_cache()
    _, ls_amp, g_amp = _oracle_grads(sd, P_f, lab, y, torch.float32, autocast=True)
    torch.cuda.empty_cache()

    model = _full_width_model(sd)

    def run(bwd):
        protnote_amd.set_math_mode(math_mode)
        protnote_amd.set_backward_math(bwd)
        try:
            for p in model.parameters():
                p.grad = None
            logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
            loss = BCEWithLogitsLoss()(logits, y.to(DEV))
            loss.backward()
            return logits.detach().clone(), loss.item(), {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}
        finally:
            protnote_amd.set_backward_math("same")
            protnote_amd.set_math_mode("f32")

    lg0, l0, g0 = run("same")
    lg1, l1, g1 = run("bf16")
    assert torch.equal(lg0, lg1) and l0 == l1                      # (i) the forward is not touched
    assert (lg1.double().cpu() - lg64).abs().max().item() < 1e-3   # and stays inside the north-star bound
    np.testing.assert_allclose(l1, ls64, rtol=1e-4)

    def rel(a, ref):
        return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)

    report, changed = [], 0
    for name, ref in g64.items():
        e_same, e_bf16, e_amp = rel(g0[name], ref), rel(g1[name], ref), rel(g_amp[name], ref)
        report.append((name, e_same, e_bf16, e_amp))
        assert e_bf16 <= 2.0 * e_amp + 1e-6, (name, e_bf16, e_amp)   # (ii) AMP class, measured
        assert e_bf16 < 2e-2, (name, e_bf16)
        changed += int(rel(g1[name], g0[name]) > 1e-6)
    # (iii) everything downstream of a single-product GEMM moved; dw_out / db_out and the top BatchNorm's dgamma / dbeta come
    # from reductions in front of the first GEMM of the backward
    assert changed >= len(g64) - 4, changed
    worst = max(report, key=lambda r: r[2])
    print(f"[{math_mode}] backward bf16: worst gradient error vs f64 {worst[2]:.2e} ({worst[0]}; default backward "
          f"{worst[1]:.2e}, torch autocast(bf16) {worst[3]:.2e}); median ratio bf16-backward / autocast = "
          f"{float(np.median([r[2] / max(r[3], 1e-30) for r in report])):.3f}")


def test_backward_bf16_gemm_error_class(bwd_bf16):
    """The mode reaches only the pair head's hidden-layer backward: the public GEMM entry points keep their arithmetic."""
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(2)
    R, M, N = 65536, 256, 256
    A = torch.randn(R, M, generator=g).to(DEV)
    Bm = torch.randn(R, N, generator=g).to(DEV)
    C = torch.empty(M, N, device=DEV)
    ws = torch.empty(16 * M * N * 4 + 1024, dtype=torch.uint8, device=DEV)
    L.check(L.lib().pn_gemm_tn(L.ptr(A), M, L.ptr(Bm), N, L.ptr(C), N, R, M, N, L.ptr(ws), ws.numel(), L.stream_ptr()))
    torch.cuda.synchronize()
    ref = A.double().T @ Bm.double()
    scale = A.double().abs().T @ Bm.double().abs()
    assert ((C.double() - ref).abs() / scale).max().item() < 1e-6   # f32 class


def test_backward_bf16_training_tracks_f32():
    """200 optimisation steps (fwd + bwd + clip + Adam) of the full-width head on a 64 x 1100 pair grid from the same
    initial state, once with the default backward and once with the bf16 backward: both learn, and the loss
    trajectories stay together."""
    import protnote_amd
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import BCEWithLogitsLoss
    from protnote_amd.utils.optim import FusedClipAdam

    gen = torch.Generator().manual_seed(6)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 64, 1100
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    # learnable targets: a fixed random bilinear rule of the inputs
    U = torch.randn(1100, 1024, generator=gen).to(DEV) / 1100 ** 0.5
    y = ((P_f @ U @ lab.T) > 32.0).float()  # (the bilinear form has standard deviation sqrt(1024) = 32: ~16 % positives)
    assert 0.02 < y.mean().item() < 0.4

    def run(bwd):
        protnote_amd.set_backward_math(bwd)
        try:
            model = _full_width_model(sd)
            opt = FusedClipAdam(head_parameters(model), lr=1e-4, max_norm=1.0)
            losses = []
            for _ in range(200):
                logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
                l = BCEWithLogitsLoss()(logits, y)
                l.backward()
                opt.step()
                opt.zero_grad()
                losses.append(l.item())
            return np.array(losses)
        finally:
            protnote_amd.set_backward_math("same")

    l32, lbf = run("same"), run("bf16")
    assert l32[-1] < 0.5 * l32[0] and lbf[-1] < 0.5 * lbf[0], (l32[[0, -1]], lbf[[0, -1]])
    dev = np.abs(lbf - l32) / np.maximum(l32, 1e-3)
    print(f"200 steps: loss {l32[0]:.4f} -> {l32[-1]:.4f} (default backward), {lbf[0]:.4f} -> {lbf[-1]:.4f} (bf16 backward); "
          f"max relative gap {dev.max():.3f}, at step 50 {dev[50]:.4f}")
    assert lbf[0] == l32[0]
    assert dev[:50].max() < 0.05 and dev.max() < 0.25, (dev[:50].max(), dev.max())


def test_backward_bf16_kernel_variants_agree():
    """pn_set_bwd_deep (bit mask): bit 0 = dh = dz W on the deep-pipelined kernel (gemm_bf16.hpp: two register sets for dz,
    three LDS buffers for the weight plane, every wait a counted vmcnt), bit 1 = dW = dz^T h on the transpose-read kernel
    (16-byte row loads, K-major LDS image, ds_read_b64_tr_b16); 0 = the single-product
    instantiations of the bf16x3 kernels.  All of them put the same bf16 values into the same products in the same order:
    EVERY gradient is bit-identical across the masks 0 / 1 / 3.  Bit 2 (mask 7, the default) stores dz as bf16: see below.  Grid with a ragged last row tile, several backward chunks and an
    odd slab count per split."""
    import protnote_amd
    from protnote_amd import _lib as L
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(41)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 96, 730          # 70 080 pair rows (B % 32 == 0: the pair-sum operand of the transpose-read kernel), not a
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)   # multiple of 256; chunks of 300 labels = 28 800 rows (112.5 row tiles)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.1).float().to(DEV)
    model = _full_width_model(sd)
    model.pair_label_chunk = 300
    names = [n for n, _ in model.named_parameters()]

    def run(mask):
        L.check(L.lib().pn_set_bwd_deep(mask))
        for p in model.parameters():
            p.grad = None
        logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        BCEWithLogitsLoss()(logits, y).backward()
        return [p.grad.clone() for p in model.parameters()]

    protnote_amd.set_backward_math("bf16")
    try:
        g0, g1, g3, g3b, g7, g7b = run(0), run(1), run(3), run(3), run(7), run(7)
    finally:
        L.lib().pn_set_bwd_deep(7)
        protnote_amd.set_backward_math("same")
    assert all(float(g.abs().max()) > 0 for g in g0)
    for n, a, b, c, d in zip(names, g0, g1, g3, g3b):
        assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d), n
    # bit 2: dz stored as bf16 in place (bwd_bf16_dz.hpp), both operands of dh = dz W by LDS-DMA with natural k order inside
    # a 16-k MFMA step (the other kernels pair k = {4g..4g+3, 16+4g..}): the same bf16 values meet the same products, so the
    # top layer's weight gradient - whose dz does not pass through a dh GEMM - is bit-identical, the rest agrees to f32
    # summation order; run to run bit-reproducible
    worst = 0.0
    for n, a, b, c in zip(names, g3, g7, g7b):
        assert torch.equal(b, c), n
        if n == "output_layer.8.weight":
            assert torch.equal(a, b), n
        rel = (a - b).norm().item() / max(a.norm().item(), 1e-30)
        worst = max(worst, rel)
        assert rel < 1e-4, (n, rel)
    assert worst > 0.0   # i.e. the stored-bf16 path really ran
    print(f"bf16-stored dz vs staging-time rounding: worst gradient difference {worst:.2e}")


def test_backward_bf16_full_size_step():
    """BASELINE configs[2] size (B = 256, L = 512, N_L = 32 102, full width, bf16x3 forward): the bf16 backward next to the
    default one on the same forward - logits and loss bit-identical over all 8.2 M pairs, every gradient finite and within
    2e-2 (Frobenius) of the default backward's (the class two f32 runs of this step differ by: ReLU-mask flips over
    2.5e10 activations), and the step bit-reproducible run to run."""
    import protnote_amd
    from bench import build_model, synthetic_batch
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    dev = torch.device(DEV)
    model = build_model(dev, unit_scale_weights=True)
    model.label_embedding_noising_alpha = 0.0
    model.train()
    batch = synthetic_batch(256, 512, 32102, dev, seed=5)
    bn = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}

    def run(bwd):
        protnote_amd.set_math_mode("bf16x3")
        protnote_amd.set_backward_math(bwd)
        try:
            model.load_state_dict(bn, strict=False)
            for p in model.parameters():
                p.grad = None
            logits, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                              label_embeddings=batch["label_embeddings"])
            loss = BCEWithLogitsLoss()(logits, batch["label_multihots"])
            loss.backward()
            return (loss.item(), logits.detach().clone(),
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            protnote_amd.set_backward_math("same")
            protnote_amd.set_math_mode("f32")

    l0, lg0, g0 = run("same")
    l1, lg1, g1 = run("bf16")
    l2, lg2, g2 = run("bf16")
    assert lg0.abs().max().item() > 1.0
    assert torch.equal(lg0, lg1) and l0 == l1
    assert torch.equal(lg1, lg2) and all(torch.equal(g1[n], g2[n]) for n in g1)   # bit-reproducible
    worst = 0.0
    for n in g0:
        assert torch.isfinite(g1[n]).all(), n
        worst = max(worst, (g1[n] - g0[n]).norm().item() / max(g0[n].norm().item(), 1e-30))
    assert 1e-6 < worst < 2e-2, worst
    print(f"full size: bf16 backward vs default backward, worst gradient difference {worst:.2e} (Frobenius)")


@pytest.mark.parametrize("B,NL,latent,scale,nl", [
    (64, 1100, 512, 2, 2),    # h = 1024, two hidden layers: 4 x 4 tile grids
    (96, 700, 256, 3, 3),     # h = 768: tile grid 3 x 3 (no XCD regions)
    (72, 920, 1024, 3, 3),    # release width, B % 32 != 0: the pair-sum layer keeps dz in f32 and takes the bf16x3 kernel's
                              # single-product instantiation, the layer above it stores dz as bf16 (R % 32 == 0)
    (66, 1000, 1024, 3, 3),   # R % 32 != 0 and B % 8 != 0: no layer stores bf16, the pair-sum weight gradient stays on f32
])
def test_backward_bf16_other_widths_vs_oracle(bwd_bf16, B, NL, latent, scale, nl):
    """Widths other than the 3072 of the release config (the single-product kernels take any hidden width that is a multiple
    of 256: other tile grids, other XCD orders, no region tasks) and grids on which only some - or none - of the dedicated
    kernels apply: logits, loss and every gradient against the f64 oracle."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(17)
    h = latent * scale
    sd = random_head_sd(gen, 1100, 1024, latent, h, 2, h, nl)
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()
    lg64, ls64, g64 = _oracle_grads(sd, P_f, lab, y, torch.float64)
    torch.cuda.empty_cache()
    model = ProtNote(latent_dim=latent, output_mlp_hidden_dim_scale_factor=scale, output_mlp_num_layers=nl,
                     projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=scale)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    loss = BCEWithLogitsLoss()(logits, y.to(DEV))
    loss.backward()
    assert (logits.detach().double().cpu() - lg64).abs().max().item() < 5e-4
    np.testing.assert_allclose(loss.item(), ls64, rtol=1e-4)
    for name, p in model.named_parameters():
        ref = g64[name]
        rel = (p.grad.double().cpu() - ref).norm().item() / max(ref.norm().item(), 1e-30)
        assert rel < 1e-2, (name, rel)


@pytest.mark.parametrize("fusion", ["concatenation_diff", "concatenation_prod"])
def test_backward_bf16_other_fusions_vs_oracle(bwd_bf16, fusion):
    """FEATURE_FUSION: concatenation_diff (effective first-layer weights) and concatenation_prod (the first layer is a stored
    pair-grid GEMM too: the layer above it takes relu(bn(z1)) as its activation, no pair sum anywhere) under the bf16
    backward, full width, 64 x 1100 pairs: logits, loss and every gradient against the f64 oracle."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(23)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3, in_mult=3)
    B, NL = 64, 1100
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()
    lg64, ls64, g64 = _oracle_grads(sd, P_f, lab, y, torch.float64, fusion=fusion)
    torch.cuda.empty_cache()
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, feature_fusion=fusion)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    loss = BCEWithLogitsLoss()(logits, y.to(DEV))
    loss.backward()
    assert (logits.detach().double().cpu() - lg64).abs().max().item() < 1e-3
    np.testing.assert_allclose(loss.item(), ls64, rtol=1e-4)
    for name, p in model.named_parameters():
        ref = g64[name]
        rel = (p.grad.double().cpu() - ref).norm().item() / max(ref.norm().item(), 1e-30)
        assert rel < 1e-2, (name, rel)
