"""Depth of the output MLP at the REAL width (d = 1024, h = 3072): what the f32 error budget does as OUTPUT_MLP_NUM_LAYERS
grows (the full-size f32 logit error is 3.0e-4 at 3 hidden layers - a k-ordered f32-MFMA chain over
K = 3072 per layer - and nothing measured how fast deeper heads eat the 1e-3 bound).  Reference: get_mlp takes any depth
(ProtNote.py:337-378); the ABI advertises PN_MAX_LAYERS = 8."""
import numpy as np
import pytest
import torch

from tests.helpers import random_head_sd
from tests.test_hip_bwd_bf16 import _oracle_grads
from tests.test_hip_fwd_bf16 import _oracle_eval, _rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("depth", [3, 5, 8])
def test_real_width_head_error_vs_depth(depth):
    """B = 8 proteins (the reference's per-GPU batch, configs/base_config.yaml:7-9) x 2 000 labels, 4-layer projections,
    `depth` hidden layers of width 3072, unit-scale weights (O(1) logits): eval logits and a train step (logits, loss, every
    gradient) of the f32 and the bf16x3 mode against the float64 oracle, with torch's own f32 run of the oracle as yardstick.
    Bars: logits inside the north-star bound 1e-3 in both modes at every depth; gradients within 4 x torch-f32's error (the
    criterion of test_train_real_width_vs_oracle).  Prints the error-vs-depth line DESIGN 7 quotes."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    gen = torch.Generator().manual_seed(60 + depth)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, depth)
    B, NL = 8, 2000
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.1).float()
    lg64, ls64, g64 = _oracle_grads(sd, P_f, lab, y, torch.float64)
    lg32, _, g32 = _oracle_grads(sd, P_f, lab, y, torch.float32)
    ev64 = _oracle_eval(sd, P_f, lab, torch.float64)
    ev32 = _oracle_eval(sd, P_f, lab, torch.float32)
    torch.cuda.empty_cache()
    t32_train, t32_eval = (lg32 - lg64).abs().max().item(), (ev32 - ev64).abs().max().item()
    f32_err = {n: _rel(g32[n], g64[n]) for n in g64}
    assert lg64.abs().max().item() > 1.0 and ev64.abs().max().item() > 1.0

    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=depth, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV)
    line = []
    for mode in ("f32", "bf16x3"):
        model.math_mode = mode
        model.load_state_dict(sd)
        model.eval()
        with torch.no_grad():
            ev, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
        e_eval = (ev.double().cpu() - ev64).abs().max().item()
        model.train()
        for p in model.parameters():
            p.grad = None
        logits, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
        loss = BCEWithLogitsLoss()(logits, y.to(DEV))
        loss.backward()
        e_train = (logits.detach().double().cpu() - lg64).abs().max().item()
        np.testing.assert_allclose(loss.item(), ls64, rtol=1e-4)
        worst = ("", 0.0, 0.0)
        for n, p in model.named_parameters():
            rel = _rel(p.grad.double().cpu(), g64[n])
            ratio = rel / max(f32_err[n], 1e-30)
            if ratio > worst[2]:
                worst = (n, rel, ratio)
            assert rel < max(4.0 * f32_err[n], 1e-6) and rel < 2e-2, (mode, depth, n, rel, f32_err[n])
        line.append(f"{mode}: eval {e_eval:.2e} train {e_train:.2e} (worst gradient {worst[0]} {worst[1]:.2e} = {worst[2]:.2f} x torch-f32)")
        assert e_eval < 1e-3 and e_train < 1e-3, (mode, depth, e_eval, e_train, t32_eval, t32_train)
    print(f"depth {depth} (8 x 2000 pairs, h = 3072): max |logit - f64|  " + ";  ".join(line) +
          f";  torch-f32: eval {t32_eval:.2e} train {t32_train:.2e}")
