"""forward_math = "bf16" (pn_set_forward_math / pn_pairhead.forward_math / model.forward_math): the hidden layers' FORWARD
pair-grid GEMMs (z_l = h_{l-1} W_l^T, l >= 1) on ONE bf16 product with f32 accumulation - the arithmetic class of the
reference's own GPU run, whose forward executes under torch.autocast (ProtNoteTrainer.py:287 eval, :728-729 train).  Together
with backward_math = "bf16" (tests/test_hip_bwd_bf16.py) this completes the reference's mixed-precision class for the output
MLP.  Opt-in and never the headline: rounding the h x h weights to bf16 alone moves O(1) logits by ~1e-2, so the yardstick
here is torch's own autocast(bfloat16) run of the oracle's formulation against the float64 oracle, not the 1e-3 bound."""
import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import random_head_sd
from tests.test_hip_bwd_bf16 import _full_width_model, _oracle_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_eval(sd, P_f, lab, dtype, autocast=False, fusion="concatenation", ndesc=1):
    """Eval-mode logits of the oracle's naive formulation run by stock torch on the device."""
    ref_sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()).to(DEV) for k, v in sd.items()}
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else torch.autocast("cuda", enabled=False)
    with torch.no_grad(), ctx:
        lg = O.protnote_forward(ref_sd, None, None, lab.to(dtype).to(DEV), training=False, fusion=fusion,
                                sequence_embeddings=P_f.to(dtype).to(DEV), descriptions_per_label=ndesc)
    return lg.double().cpu()


def _rel(a, ref):
    return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)


@pytest.mark.parametrize("math_mode,bwd", [("f32", "same"), ("bf16x3", "same"), ("bf16x3", "bf16"), ("f32", "bf16")])
def test_forward_bf16_step_vs_f64_with_autocast_yardstick(math_mode, bwd):
    """Full-width head, 256 x 300 pairs, train mode (batch-statistics BatchNorm over the grid).  Against the float64 oracle:
    (i) the logits of forward_math = "bf16" are within 2 x the error torch's own autocast(bfloat16) run of the oracle shows
    (max and rms), and NOT equal to forward_math = "same" (the single-product kernels really ran); (ii) the loss agrees to
    2e-3 relative; (iii) every gradient is within 2 x torch-autocast's error (the AMP criterion of the bf16 backward),
    absolute cap 5e-2 - with the default backward behind it as well as with the bf16 backward."""
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
    lg_amp, ls_amp, g_amp = _oracle_grads(sd, P_f, lab, y, torch.float32, autocast=True)
    torch.cuda.empty_cache()
    amp_max = (lg_amp - lg64).abs().max().item()
    amp_rms = (lg_amp - lg64).pow(2).mean().sqrt().item()

    model = _full_width_model(sd)

    def run(fwd):
        model.math_mode, model.forward_math, model.backward_math = math_mode, fwd, bwd
        for p in model.parameters():
            p.grad = None
        logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
        loss = BCEWithLogitsLoss()(logits, y.to(DEV))
        loss.backward()
        return logits.detach().double().cpu(), loss.item(), {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}

    lg0, l0, g0 = run("same")
    lg1, l1, g1 = run("bf16")
    lg2, l2, g2 = run("bf16")
    assert torch.equal(lg1, lg2) and l1 == l2 and all(torch.equal(g1[n], g2[n]) for n in g1)   # bit-reproducible
    moved = (lg1 - lg0).abs().max().item()
    assert moved > 1e-4, moved                                    # the single-product kernels ran
    assert (lg0 - lg64).abs().max().item() < 1e-3                 # (the default forward stays inside the north-star bound)
    e_max = (lg1 - lg64).abs().max().item()
    e_rms = (lg1 - lg64).pow(2).mean().sqrt().item()
    assert e_max <= 2.0 * amp_max and e_rms <= 2.0 * amp_rms, (e_max, amp_max, e_rms, amp_rms)   # (i)
    assert abs(l1 - ls64) <= 2e-3 * abs(ls64), (l1, ls64, ls_amp)                               # (ii)
    report = []
    for name, ref in g64.items():
        e_same, e_bf16, e_amp = _rel(g0[name], ref), _rel(g1[name], ref), _rel(g_amp[name], ref)
        report.append((name, e_same, e_bf16, e_amp))
        assert e_bf16 <= 2.0 * e_amp + 1e-6, (name, e_bf16, e_amp)                              # (iii)
        assert e_bf16 < 5e-2, (name, e_bf16)
    worst = max(report, key=lambda r: r[2])
    print(f"[{math_mode}, forward bf16, backward {bwd}] logits vs f64: max {e_max:.2e} rms {e_rms:.2e} (torch autocast(bf16): max "
          f"{amp_max:.2e} rms {amp_rms:.2e}; forward 'same': max {(lg0 - lg64).abs().max().item():.2e}); loss {l1:.6f} vs f64 "
          f"{ls64:.6f} (autocast {ls_amp:.6f}); worst gradient {worst[0]}: {worst[2]:.2e} (forward 'same' {worst[1]:.2e}, "
          f"autocast {worst[3]:.2e}); median ratio to autocast {float(np.median([r[2] / max(r[3], 1e-30) for r in report])):.3f}")


@pytest.mark.parametrize("ndesc", [1, 2])
def test_forward_bf16_eval_vs_f64_with_autocast_yardstick(ndesc):
    """Eval mode (running-statistics BatchNorm, the fused inference kernels under no_grad, description ensembling): logits of
    forward_math = "bf16" within 2 x torch-autocast's error against the float64 oracle, on the f32 and the bf16x3 base mode."""
    gen = torch.Generator().manual_seed(35)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 256, 300
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = _oracle_eval(sd, P_f, lab, torch.float64, ndesc=ndesc)
    amp = _oracle_eval(sd, P_f, lab, torch.float32, autocast=True, ndesc=ndesc)
    amp_max, amp_rms = (amp - ref).abs().max().item(), (amp - ref).pow(2).mean().sqrt().item()
    model = _full_width_model(sd).eval()
    model.inference_descriptions_per_label = ndesc
    for math_mode in ("f32", "bf16x3"):
        model.math_mode = math_mode
        out = {}
        for fwd in ("same", "bf16"):
            model.forward_math = fwd
            with torch.no_grad():
                lg, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
            out[fwd] = lg.double().cpu()
        assert out["bf16"].shape == (B, NL // ndesc)
        assert (out["same"] - ref).abs().max().item() < 1e-3
        e_max = (out["bf16"] - ref).abs().max().item()
        e_rms = (out["bf16"] - ref).pow(2).mean().sqrt().item()
        assert (out["bf16"] - out["same"]).abs().max().item() > 1e-4
        assert e_max <= 2.0 * amp_max and e_rms <= 2.0 * amp_rms, (math_mode, e_max, amp_max, e_rms, amp_rms)
        print(f"[eval, {math_mode}, ndesc {ndesc}] forward bf16 logits vs f64: max {e_max:.2e} rms {e_rms:.2e} "
              f"(torch autocast(bf16): max {amp_max:.2e} rms {amp_rms:.2e})")


def test_forward_bf16_map_parity():
    """BASELINE metric 'mAP parity vs reference' with the AMP-class forward: micro / macro AP of the device metrics over
    the HIP logits (full-width head, 256 x 300 pairs) against the AP of the CPU oracle's f32 logits and against the f32 HIP
    path.  The reference's own seed-to-seed spread is mAP-micro +- 0.0013: |delta mAP| must stay below it."""
    from oracle import metrics_oracle as MO
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    gen = torch.Generator().manual_seed(31)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 256, 300
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f)
    y = (torch.rand(B, NL, generator=gen) < torch.sigmoid(2 * ref - 2)).numpy()
    model = _full_width_model(sd).eval()
    got = {}
    for fwd in ("same", "bf16"):
        model.forward_math = fwd
        with torch.no_grad():
            out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
        acc = DeviceAveragePrecision(NL, B, DEV)
        acc.update(torch.sigmoid(out), torch.from_numpy(y).to(DEV))
        got[fwd] = acc.compute()
    pr = torch.sigmoid(ref).numpy()
    mi_ref = MO.average_precision_fast(pr.ravel(), y.ravel())
    ma_ref = MO.macro_mean([MO.average_precision_fast(pr[:, j], y[:, j]) for j in range(NL)])
    assert 0.2 < mi_ref < 0.99
    d_mi = abs(got["bf16"]["map_micro"] - mi_ref)
    d_ma = abs(got["bf16"]["map_macro"] - ma_ref)
    print(f"mAP parity, forward bf16: micro {got['bf16']['map_micro']:.6f} (oracle {mi_ref:.6f}, f32 path "
          f"{got['same']['map_micro']:.6f}; |delta| {d_mi:.2e}), macro {got['bf16']['map_macro']:.6f} (oracle {ma_ref:.6f}; "
          f"|delta| {d_ma:.2e}); reference seed spread 1.3e-3")
    assert d_mi < 1.3e-3 and d_ma < 2.6e-3, (got, mi_ref, ma_ref)


def test_forward_bf16_training_tracks_f32():
    """200 optimisation steps (fwd + bwd + clip + Adam) of the full-width head on a 64 x 1100 pair grid from the same
    initial state: default arithmetic vs the full AMP class (forward bf16 + backward bf16).  Both learn; the loss
    trajectories stay together."""
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.losses import BCEWithLogitsLoss
    from protnote_amd.utils.optim import FusedClipAdam

    gen = torch.Generator().manual_seed(6)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 64, 1100
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    U = torch.randn(1100, 1024, generator=gen).to(DEV) / 1100 ** 0.5
    y = ((P_f @ U @ lab.T) > 32.0).float()
    assert 0.02 < y.mean().item() < 0.4

    def run(fwd, bwd):
        model = _full_width_model(sd)
        model.forward_math, model.backward_math = fwd, bwd
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

    l32, lamp = run("same", "same"), run("bf16", "bf16")
    assert l32[-1] < 0.5 * l32[0] and lamp[-1] < 0.5 * lamp[0], (l32[[0, -1]], lamp[[0, -1]])
    dev = np.abs(lamp - l32) / np.maximum(l32, 1e-3)
    print(f"200 steps: loss {l32[0]:.4f} -> {l32[-1]:.4f} (default), {lamp[0]:.4f} -> {lamp[-1]:.4f} (forward + backward bf16); "
          f"max relative gap {dev.max():.3f}, first step {dev[0]:.2e}, at step 50 {dev[50]:.4f}")
    assert dev[0] < 5e-3
    assert dev[:50].max() < 0.05 and dev.max() < 0.25, (dev[:50].max(), dev.max())


@pytest.mark.parametrize("B,NL,latent,scale,nl,fusion", [
    (64, 1100, 512, 2, 2, "concatenation"),        # h = 1024, two hidden layers: ONE pair-grid GEMM, 4 column tiles
    (96, 700, 256, 3, 3, "concatenation"),         # h = 768: 3 column tiles (no XCD regions)
    (8, 333, 1024, 3, 3, "concatenation"),         # the reference's per-GPU batch: 2 664 pair rows = 10.4 row tiles
    (64, 1100, 1024, 3, 3, "concatenation_diff"),  # effective first-layer weights
    (64, 1100, 1024, 3, 3, "concatenation_prod"),  # z1 is a stored pair-grid GEMM (stays math_mode); layers 2, 3 take relu(bn(z))
    (66, 1000, 1024, 3, 5, "concatenation"),       # five hidden layers: four single-product GEMMs in a row
])
def test_forward_bf16_other_shapes_vs_oracle(B, NL, latent, scale, nl, fusion):
    """Other widths (any hidden width that is a multiple of 256), depths, fusions and grids: train step with the full AMP
    class and eval forward against the f64 oracle, held to the same yardstick as the full-width test - logits (max and rms)
    and every gradient within 2 x the error of torch's own autocast(bfloat16) run of the oracle on the same case - and every
    case is checked to have really left the f32 path."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(17)
    h = latent * scale
    in_mult = 2 if fusion == "concatenation" else 3
    sd = random_head_sd(gen, 1100, 1024, latent, h, 2, h, nl, in_mult=in_mult)
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()
    lg64, ls64, g64 = _oracle_grads(sd, P_f, lab, y, torch.float64, fusion=fusion)
    lg_amp, _, g_amp = _oracle_grads(sd, P_f, lab, y, torch.float32, autocast=True, fusion=fusion)
    ev64 = _oracle_eval(sd, P_f, lab, torch.float64, fusion=fusion)
    ev_amp = _oracle_eval(sd, P_f, lab, torch.float32, autocast=True, fusion=fusion)
    torch.cuda.empty_cache()

    def stats(x, ref):
        return (x - ref).abs().max().item(), (x - ref).pow(2).mean().sqrt().item()

    model = ProtNote(latent_dim=latent, output_mlp_hidden_dim_scale_factor=scale, output_mlp_num_layers=nl,
                     projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=scale, feature_fusion=fusion)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    model.forward_math, model.backward_math = "bf16", "bf16"
    logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    loss = BCEWithLogitsLoss()(logits, y.to(DEV))
    loss.backward()
    (e_max, e_rms), (a_max, a_rms) = stats(logits.detach().double().cpu(), lg64), stats(lg_amp, lg64)
    assert e_max > 1e-5 and e_max <= 2.0 * a_max and e_rms <= 2.0 * a_rms, (e_max, a_max, e_rms, a_rms)
    np.testing.assert_allclose(loss.item(), ls64, rtol=5e-3)
    worst = ("", 0.0, 0.0)
    for name, p in model.named_parameters():
        rel, amp = _rel(p.grad.double().cpu(), g64[name]), _rel(g_amp[name], g64[name])
        if rel > worst[1]:
            worst = (name, rel, amp)
        assert rel <= 2.0 * amp + 1e-6 and rel < 0.25, (name, rel, amp)   # (five bf16 layers: 0.10 on W_l.0, autocast 0.16)
    model.load_state_dict(sd)  # (the train-mode forward advanced the BatchNorm buffers)
    model.eval()
    with torch.no_grad():
        ev, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    (v_max, v_rms), (va_max, va_rms) = stats(ev.double().cpu(), ev64), stats(ev_amp, ev64)
    assert v_max > 1e-5 and v_max <= 2.0 * va_max and v_rms <= 2.0 * va_rms, (v_max, va_max, v_rms, va_rms)
    print(f"[{fusion}, {B} x {NL}, h = {h}, {nl} hidden layers] forward + backward bf16 vs f64: train logits max {e_max:.2e} rms "
          f"{e_rms:.2e} (autocast {a_max:.2e} / {a_rms:.2e}), eval logits max {v_max:.2e} rms {v_rms:.2e} (autocast {va_max:.2e} "
          f"/ {va_rms:.2e}), worst gradient {worst[0]} {worst[1]:.2e} (autocast {worst[2]:.2e})")


def test_forward_bf16_falls_back_where_the_kernel_does_not_apply():
    """Hidden width not a multiple of 256 (h = 600): the single-product kernel does not cover the shape, the layer keeps
    math_mode's kernels - logits bit-identical to forward_math = "same".  Likewise with OUTPUT_MLP_DROPOUT > 0 (the dropped
    layers stay on the kernels that carry the mask code): same seed, same logits."""
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(3)
    sd = random_head_sd(gen, 1100, 1024, 200, 600, 2, 600, 3)
    P_f = torch.randn(40, 1100, generator=gen).to(DEV)
    lab = torch.randn(90, 1024, generator=gen).to(DEV)
    model = ProtNote(latent_dim=200, output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3,
                     projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    outs = {}
    for fwd in ("same", "bf16"):
        model.forward_math = fwd
        with torch.no_grad():
            outs[fwd], _ = model(sequence_embeddings=P_f, label_embeddings=lab)
    assert torch.equal(outs["same"], outs["bf16"])

    sd = random_head_sd(gen, 1100, 1024, 256, 768, 2, 768, 3)
    model = ProtNote(latent_dim=256, output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3,
                     projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=3, dropout=0.1)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    outs = {}
    for fwd in ("same", "bf16"):
        model.load_state_dict(sd)
        model.forward_math = fwd
        torch.manual_seed(123)
        with torch.no_grad():
            outs[fwd], _ = model(sequence_embeddings=P_f, label_embeddings=lab)
    assert torch.equal(outs["same"], outs["bf16"])


@pytest.mark.parametrize("training", [False, True])
def test_forward_bf16_saved_embeddings_belong_to_the_logits(training):
    """save_embeddings=True (ProtNote.py:292-302) under forward_math = "bf16": the penultimate activations that come back are
    the ones the logits were computed from (logit = hidden . w_out + b_out to f32 rounding), in eval mode (fused kernels +
    pn_pairhead_fwd_eval_hidden) and in train mode (activation store)."""
    gen = torch.Generator().manual_seed(12)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 24, 40
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    model = _full_width_model(sd)
    model.train(training)
    res = {}
    for fwd in ("same", "bf16"):
        model.load_state_dict(sd)
        model.forward_math = fwd
        with torch.no_grad():
            lg, emb = model(sequence_embeddings=P_f, label_embeddings=lab, save_embeddings=True)
        hid = emb["output_layer_embeddings"].double()
        assert hid.shape == (B * NL, 3072)
        w = sd["output_layer.11.weight"].double().reshape(-1)
        b = sd["output_layer.11.bias"].double().item()
        res[fwd] = (lg.double().cpu().reshape(-1), hid @ w + b, hid)
        assert (res[fwd][0] - res[fwd][1]).abs().max().item() < 1e-4, fwd
    assert (res["bf16"][2] - res["same"][2]).abs().max().item() > 1e-4  # and they are the bf16 forward's activations


def test_forward_math_switches():
    """The three ways to select the mode - process default (set_forward_math / PN_FORWARD_MATH), per model
    (model.forward_math), per call (descriptor field) - and their error behaviour."""
    import protnote_amd
    from protnote_amd import _lib as L

    assert protnote_amd.get_forward_math() == "same"
    with pytest.raises(ValueError):
        protnote_amd.set_forward_math("fp8")
    assert L.lib().pn_set_forward_math(2) != 0 and b"pn_set_forward_math" in L.lib().pn_last_error()
    gen = torch.Generator().manual_seed(5)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    P_f = torch.randn(16, 1100, generator=gen).to(DEV)
    lab = torch.randn(50, 1024, generator=gen).to(DEV)
    model = _full_width_model(sd).eval()

    def fwd():
        with torch.no_grad():
            return model(sequence_embeddings=P_f, label_embeddings=lab)[0]

    base = fwd()
    model.forward_math = "bf16"
    per_model = fwd()
    model.forward_math = None
    protnote_amd.set_forward_math("bf16")
    try:
        assert protnote_amd.get_forward_math() == "bf16"
        by_default = fwd()
        model.forward_math = "same"   # an explicit per-model choice beats the process default
        explicit_same = fwd()
    finally:
        protnote_amd.set_forward_math("same")
        model.forward_math = None
    assert torch.equal(per_model, by_default) and not torch.equal(per_model, base)
    assert torch.equal(explicit_same, base)


@pytest.mark.parametrize("fusion,nl,B,NL", [("concatenation", 3, 96, 730), ("concatenation_prod", 3, 64, 500),
                                             ("concatenation", 2, 40, 300), ("concatenation_diff", 5, 33, 410)])
def test_forward_bf16_routes_agree(fusion, nl, B, NL):
    """pn_set_fwd_staged: 1 (default) = the activation operand is written once per chunk as bf16 and both operands go by
    LDS-DMA (fwd_bf16_h.hpp: k_make_h_bf16 + gemm_nt_bf16dma_kernel, E_STORE_H16 producers in eval); 0 = the register-staged
    single-product kernels round it while staging.  The same bf16 values meet in the same products; what differs is the k order
    inside a 16-k MFMA step, i.e. the f32 rounding of every pre-activation - and a pre-activation that moves by one f32 ulp can
    land on the other side of a bf16 rounding boundary of the NEXT layer's operand (a few elements per row and layer, each
    worth one bf16 ulp).  So the routes are two equally good draws of the same arithmetic class, not bit-twins: held here are
    (i) each route is bit-reproducible, (ii) both are the same distance from the float64 oracle (eval and train logits within
    1.25 x of each other), (iii) they are closer to each other than to the oracle, (iv) BatchNorm buffers and gradients
    agree at the bf16 class."""
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(51)
    in_mult = 2 if fusion == "concatenation" else 3
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, nl, in_mult=in_mult)
    P_c = torch.randn(B, 1100, generator=gen)
    lab_c = torch.randn(NL, 1024, generator=gen)
    y_c = (torch.rand(B, NL, generator=gen) < 0.1).float()
    lg64, _, g64 = _oracle_grads(sd, P_c, lab_c, y_c, torch.float64, fusion=fusion)
    ev64 = _oracle_eval(sd, P_c, lab_c, torch.float64, fusion=fusion)
    torch.cuda.empty_cache()
    P_f, lab, y = P_c.to(DEV), lab_c.to(DEV), y_c.to(DEV)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=nl, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, feature_fusion=fusion)
    model.load_state_dict(sd)
    model = model.to(DEV)
    model.pair_label_chunk = 300  # several chunks in the eval head

    def run(staged):
        L.check(L.lib().pn_set_fwd_staged(staged))
        try:
            model.forward_math = "bf16"
            model.load_state_dict(sd)
            model.eval()
            with torch.no_grad():
                ev, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
            model.train()
            for p in model.parameters():
                p.grad = None
            lg, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
            BCEWithLogitsLoss()(lg, y).backward()
            bufs = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
            return ev.double().cpu(), lg.detach().double().cpu(), bufs, {n: p.grad.double().cpu() for n, p in model.named_parameters()}
        finally:
            L.lib().pn_set_fwd_staged(1)

    a, a2, b = run(1), run(1), run(0)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1]) and all(torch.equal(a[3][n], a2[3][n]) for n in a[3])   # (i)
    out = []
    for what, x, y_, ref in (("eval", a[0], b[0], ev64), ("train", a[1], b[1], lg64)):
        e_st, e_rg, d = float((x - ref).abs().max()), float((y_ - ref).abs().max()), float((x - y_).abs().max())
        r_st, r_rg = float((x - ref).pow(2).mean().sqrt()), float((y_ - ref).pow(2).mean().sqrt())
        out.append(f"{what}: staged {e_st:.2e} (rms {r_st:.2e}) register-staged {e_rg:.2e} (rms {r_rg:.2e}) between the routes {d:.2e}")
        assert d > 0.0                                                                  # the other kernels really ran
        assert r_st <= 1.25 * r_rg and r_rg <= 1.25 * r_st, (what, r_st, r_rg)           # (ii)
        assert float((x - y_).pow(2).mean().sqrt()) < 0.5 * max(r_st, r_rg), (what, d)   # (iii)
    for k in a[2]:
        np.testing.assert_allclose(a[2][k].cpu().numpy(), b[2][k].cpu().numpy(), rtol=2e-3, atol=2e-4, err_msg=k)
    worst = ("", 0.0, 0.0)
    for n in a[3]:
        e_st, e_rg = _rel(a[3][n], g64[n]), _rel(b[3][n], g64[n])
        if e_st > worst[1]:
            worst = (n, e_st, e_rg)
        assert e_st <= 1.5 * e_rg + 1e-4 and e_rg <= 1.5 * e_st + 1e-4, (n, e_st, e_rg)   # (iv)
    print(f"[{fusion}, {nl} layers, {B} x {NL}] bf16 forward, logits vs f64 - " + "; ".join(out) +
          f"; worst gradient {worst[0]}: staged {worst[1]:.2e} register-staged {worst[2]:.2e}")


def test_mfma16_matches_mfma32():
    """pn_set_bf16_mfma16: the one-product bf16 GEMMs - the all-LDS-DMA NT kernel (gemm_bf16_m16.hpp) and the transpose-read
    weight-gradient kernel (gemm_bf16.hpp, M16) - issue v_mfma_f32_16x16x32_bf16 (default) or v_mfma_f32_32x32x16_bf16.  Every
    accumulator is bit-identical between the two (tools/lab_bf16_nt.hip compares z and the bf16 h on the device at
    262 144 x 3072 x 3072); what differs is the reduction order of the fused epilogues.  Held here:
    (i) backward_math = bf16 over an f32 forward - dh = dz W and dW = dz^T h are plain stores, so EVERY gradient is bit-identical;
    (ii) forward_math = bf16, eval - h is bit-identical, the logits are row dots summed in another order: f32 rounding only;
    (iii) forward_math = bf16, train - the BatchNorm column partials sum in another order, batch statistics move in the last ulp, a few
    elements of the next bf16 operand round the other way: same arithmetic class (same distance from the float64 oracle, rms of the
    difference far below it)."""
    from protnote_amd import _lib as L
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(77)
    B, NL, nl = 72, 520, 3
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, nl)
    P_c = torch.randn(B, 1100, generator=gen)
    lab_c = torch.randn(NL, 1024, generator=gen)
    y_c = (torch.rand(B, NL, generator=gen) < 0.1).float()
    lg64, _, _ = _oracle_grads(sd, P_c, lab_c, y_c, torch.float64)
    torch.cuda.empty_cache()
    P_f, lab, y = P_c.to(DEV), lab_c.to(DEV), y_c.to(DEV)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=nl, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, feature_fusion="concatenation")
    model.load_state_dict(sd)
    model = model.to(DEV)
    model.pair_label_chunk = 200

    def run(m16, fwd, bwd):
        L.check(L.lib().pn_set_bf16_mfma16(m16))
        try:
            model.forward_math, model.backward_math = fwd, bwd
            model.load_state_dict(sd)
            model.eval()
            with torch.no_grad():
                ev, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
            model.train()
            for p in model.parameters():
                p.grad = None
            lg, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
            BCEWithLogitsLoss()(lg, y).backward()
            return ev.clone(), lg.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}
        finally:
            L.lib().pn_set_bf16_mfma16(1)
            model.forward_math, model.backward_math = "same", "same"

    # (i)
    a, b = run(1, "same", "bf16"), run(0, "same", "bf16")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n in a[2]:
        assert torch.equal(a[2][n], b[2][n]), n
    # (ii), (iii)
    a, b = run(1, "bf16", "bf16"), run(0, "bf16", "bf16")
    scale = float(a[0].abs().max())
    d_ev = float((a[0] - b[0]).abs().max())
    assert d_ev <= 4e-6 * scale, (d_ev, scale)
    ref = lg64.to(DEV)
    r16 = float((a[1].double() - ref).pow(2).mean().sqrt())
    r32 = float((b[1].double() - ref).pow(2).mean().sqrt())
    d_tr = float((a[1] - b[1]).double().pow(2).mean().sqrt())
    assert r16 <= 1.05 * r32 and r32 <= 1.05 * r16, (r16, r32)
    assert d_tr <= 0.1 * max(r16, r32), (d_tr, r16, r32)
    for n in a[2]:
        g16, g32 = a[2][n].double(), b[2][n].double()
        assert float((g16 - g32).norm()) <= 2e-2 * float(g32.norm()) + 1e-12, n
    print(f"16x16x32 vs 32x32x16: gradients of an f32 forward bit-identical; bf16 forward - eval logits differ by {d_ev:.1e} at scale "
          f"{scale:.1f}; train logits rms vs f64 {r16:.2e} / {r32:.2e}, rms between the two {d_tr:.1e}")


@pytest.mark.parametrize("fusion", ["concatenation", "concatenation_diff", "concatenation_prod"])
def test_forward_bf16_one_hidden_layer_is_unchanged(fusion):
    """OUTPUT_MLP_NUM_LAYERS: 1 has no hidden pair-grid GEMM (the separable layer is the only hidden layer; concatenation_prod's
    extra layer-1 GEMM follows math_mode by definition), so forward_math = "bf16" changes nothing: eval logits, train logits and
    every gradient bit-identical to forward_math = "same" - at a width the staged route would otherwise take."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(77)
    in_mult = 2 if fusion == "concatenation" else 3
    sd = random_head_sd(gen, 1100, 1024, 256, 768, 2, 768, 1, in_mult=in_mult)
    B, NL = 24, 130
    P_f = torch.randn(B, 1100, generator=gen).to(DEV)
    lab = torch.randn(NL, 1024, generator=gen).to(DEV)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float().to(DEV)
    model = ProtNote(latent_dim=256, output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=1, projection_head_num_layers=2,
                     projection_head_hidden_dim_scale_factor=3, feature_fusion=fusion)
    model.load_state_dict(sd)
    model = model.to(DEV)
    model.pair_label_chunk = 50
    out = {}
    for fwd in ("same", "bf16"):
        model.forward_math = fwd
        model.load_state_dict(sd)
        model.eval()
        with torch.no_grad():
            ev, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        model.train()
        for p in model.parameters():
            p.grad = None
        lg, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
        BCEWithLogitsLoss()(lg, y).backward()
        out[fwd] = (ev.clone(), lg.detach().clone(), [p.grad.clone() for p in model.parameters()])
    assert torch.isfinite(out["bf16"][0]).all() and float(out["bf16"][0].abs().max()) > 0.1
    assert torch.equal(out["same"][0], out["bf16"][0]) and torch.equal(out["same"][1], out["bf16"][1])
    assert all(torch.equal(a, b) for a, b in zip(out["same"][2], out["bf16"][2]))


def test_forward_bf16_without_batchnorm_and_in_a_differentiated_eval_forward():
    """Two corners of the mode: (i) OUTPUT_MLP_BATCHNORM: False (Linear bias + ReLU; the bias rides in the fold's shift) - train
    step with forward + backward bf16 against the f64 oracle at the bf16 class; (ii) model.eval() with autograd on (the
    activation-storing path with BatchNorm on its running statistics) gives the logits of the fused inference kernels under
    no_grad to a few bf16 ulps (the stored f32 pre-activation goes through the same fold and rounding as the fused producer's
    epilogue; the row MLPs in front differ in their last f32 bits)."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(91)
    torch.manual_seed(91)
    model = ProtNote(protein_embedding_dim=64, label_embedding_dim=48, latent_dim=128, output_mlp_hidden_dim_scale_factor=2,
                     output_mlp_num_layers=3, outout_mlp_add_batchnorm=False, projection_head_num_layers=2,
                     projection_head_hidden_dim_scale_factor=2)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * (1.6 / m.weight.shape[1] ** 0.5))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
            elif isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.3)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B, NL = 12, 50
    P_f = torch.randn(B, 64, generator=gen)
    lab = torch.randn(NL, 48, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.2).float()
    lg64, ls64, g64 = _oracle_grads(sd, P_f, lab, y, torch.float64)
    model = model.to(DEV).train()
    model.forward_math, model.backward_math = "bf16", "bf16"
    lg, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    loss = BCEWithLogitsLoss()(lg, y.to(DEV))
    loss.backward()
    scale = max(1.0, float(lg64.abs().max()))
    err = float((lg.detach().double().cpu() - lg64).abs().max())
    assert 1e-5 < err < 3e-2 * scale, (err, scale)
    np.testing.assert_allclose(loss.item(), ls64, rtol=5e-3)
    for n, p in model.named_parameters():
        assert _rel(p.grad.double().cpu(), g64[n]) < 8e-2, n

    # (ii) full width, BatchNorm on: eval + autograd == eval under no_grad
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    m2 = _full_width_model(sd).eval()
    m2.forward_math = "bf16"
    P2, l2 = torch.randn(40, 1100, generator=gen).to(DEV), torch.randn(90, 1024, generator=gen).to(DEV)
    with torch.no_grad():
        fused, _ = m2(sequence_embeddings=P2, label_embeddings=l2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        stored, _ = m2(sequence_embeddings=P2, label_embeddings=l2)
    assert stored.requires_grad
    # (W_p / W_l run other kernels on the two paths - eval vs activation-storing row MLPs - so P_e, L_e differ in the last f32
    #  bits, and such a difference can cross a bf16 rounding boundary of a hidden operand: the two paths agree at a few bf16
    #  ulps of a few elements - measured 1.6e-2 at logit scale 7.9 - far inside the 7e-2 the mode itself is from f64)
    d = float((stored.detach() - fused).abs().max())
    rms = float((stored.detach() - fused).pow(2).mean().sqrt())
    assert d < 5e-3 * max(1.0, float(fused.abs().max())) and rms < 1e-3 * max(1.0, float(fused.abs().max())), (d, rms)
    stored.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m2.output_layer.parameters())
