"""GPU parity tests: the HIP path (through the C ABI) vs the CPU oracle and the reference-generated golden
vectors.  Tolerance (BASELINE.json north_star): logits within 1e-3 of the fp32 CPU path; the kernels are
exact-f32 MFMA so the tests hold them to 2e-4 abs on O(1) values."""
import os

import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import make_encoder, make_protnote, npz_cfg, random_encoder_sd, random_head_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,variant", [(128, 128, 32, 0), (300, 200, 100, 0), (257, 70, 36, 1), (1, 5, 4, 0),
                                            (513, 550, 1100, -1), (64, 3072, 1024, 0)])
def test_gemm_nt(M, N, K, variant):
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    # asymmetric W so a transposed C-write cannot pass
    W = torch.randn(N, K, generator=g) + torch.arange(N)[:, None] * 0.01
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + bias.double()
    Ad, Wd, bd = A.to(DEV), W.to(DEV), bias.to(DEV)
    Cd = torch.full((M, N), float("nan"), device=DEV)
    cs = torch.full((N,), float("nan"), dtype=torch.float64, device=DEV)  # written, not accumulated
    cq = torch.full((N,), float("nan"), dtype=torch.float64, device=DEV)
    ws = torch.empty(L.lib().pn_gemm_nt_stats_ws_bytes(M, N), dtype=torch.uint8, device=DEV)

    def run():
        L.check(L.lib().pn_gemm_nt(L.ptr(Ad), K, L.ptr(Wd), K, L.ptr(Cd), N, M, N, K, L.ptr(bd), None, None,
                                   L.ptr(cs), L.ptr(cq), variant, L.ptr(ws), ws.numel(), L.stream_ptr()))
        torch.cuda.synchronize()
        return cs.clone(), cq.clone()

    first = run()
    again = run()
    assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])  # fixed-order reduction: same bits
    out = Cd.cpu().double()
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-6 * scale * max(1, K ** 0.5)
    np.testing.assert_allclose(cs.cpu().numpy(), ref.sum(0).numpy(), rtol=1e-5, atol=1e-4 * scale)
    np.testing.assert_allclose(cq.cpu().numpy(), (ref ** 2).sum(0).numpy(), rtol=1e-5)


def test_gemm_nt_affine_relu():
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(3)
    M, N, K = 200, 96, 52
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    s, t = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    ref = torch.relu(A.double() * s.double() + t.double()) @ W.double().T
    Cd = torch.empty(M, N, device=DEV)
    Ad, Wd, sd_, td = A.to(DEV), W.to(DEV), s.to(DEV), t.to(DEV)  # keep alive across the async launch
    L.check(L.lib().pn_gemm_nt(L.ptr(Ad), K, L.ptr(Wd), K, L.ptr(Cd), N, M, N, K, None,
                               L.ptr(sd_), L.ptr(td), None, None, 0, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert (Cd.cpu().double() - ref).abs().max().item() < 2e-5


# ----------------------------------------------------------------------------------------------- encoder
def test_encoder_golden_eval_and_train(golden_dir):
    g = _g(golden_dir, "encoder_small.npz")
    sd = O.as_torch_sd(g, "sd/")
    enc = make_encoder(sd, "", npz_cfg(g, "cfg_"), DEV)
    for p in enc.parameters():
        p.requires_grad = False
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    enc.eval()
    emb = enc.get_embeddings(x, lens)
    np.testing.assert_allclose(emb.cpu().numpy(), g["eval/embeddings"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(enc(x, lens).cpu().numpy(), g["eval/logits"], atol=2e-4, rtol=1e-4)
    # pad-length invariance (bucketed padding is parity-safe): pad to 260
    x2 = torch.zeros(x.shape[0], 20, 260, device=DEV)
    x2[:, :, : x.shape[2]] = x
    np.testing.assert_allclose(enc.get_embeddings(x2, lens).cpu().numpy(), g["eval/embeddings_pad260"], atol=1e-4,
                               rtol=1e-4)
    # train-mode BN of the frozen encoder + running-stat drift
    enc.train()
    emb_t = enc.get_embeddings(x, lens)
    np.testing.assert_allclose(emb_t.cpu().numpy(), g["train/embeddings"], atol=2e-4, rtol=1e-4)
    after = O.as_torch_sd(g, "sd_after_train/")
    got = {k: v.cpu() for k, v in enc.state_dict().items()}
    for k, v in after.items():
        np.testing.assert_allclose(got[k].numpy(), v.numpy(), atol=1e-5, rtol=1e-4, err_msg=k)


def test_masked_conv1d_and_residual_standalone_golden(golden_dir):
    """The two public classes under ProteInfer, called on their own (reference protein_encoders.py:8-17, :61-67;
    pn_masked_conv1d_fwd / pn_residual_fwd) - against activations the REFERENCE produced: conv1 and block 0 of
    encoder_small.npz (one-hot input with garbage pads -> eval/conv1 -> eval/block0), and the stand-alone cases of
    encoder_pieces.npz (random input whose pads hold garbage: a dilated 52 -> 26 convolution, a 1 x 1 convolution, blocks 1 and
    4 in eval mode, block 1 in train mode with its running statistics).  What a stand-alone Residual does differently from
    the fused pipeline - statistics over the raw pads, pads passed through - is part of the fixture."""
    g = _g(golden_dir, "encoder_small.npz")
    enc = make_encoder(O.as_torch_sd(g, "sd/"), "", npz_cfg(g, "cfg_"), DEV).eval()
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    with torch.no_grad():
        c1 = enc.conv1(x, lens)
        np.testing.assert_allclose(c1.cpu().numpy(), g["eval/conv1"], atol=1e-5, rtol=1e-5)
        b0 = enc.resnet_blocks[0](c1, lens)
        np.testing.assert_allclose(b0.cpu().numpy(), g["eval/block0"], atol=2e-5, rtol=1e-5)

    g = _g(golden_dir, "encoder_pieces.npz")
    enc = make_encoder(O.as_torch_sd(g, "sd/"), "", npz_cfg(g, "cfg_"), DEV).eval()
    h, lens = torch.from_numpy(g["h"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    with torch.no_grad():
        got = enc.resnet_blocks[2].masked_conv1(h, lens)
        np.testing.assert_allclose(got.cpu().numpy(), g["conv/block2_masked_conv1"], atol=1e-5, rtol=1e-5)
        got = enc.resnet_blocks[1].masked_conv2(h[:, :26].contiguous(), lens)
        np.testing.assert_allclose(got.cpu().numpy(), g["conv/block1_masked_conv2"], atol=1e-5, rtol=1e-5)
        for blk in (1, 4):
            got = enc.resnet_blocks[blk](h, lens)
            np.testing.assert_allclose(got.cpu().numpy(), g[f"eval/residual{blk}"], atol=2e-5, rtol=1e-5)
        blk = enc.resnet_blocks[1].train()
        got = blk(h, lens)
        np.testing.assert_allclose(got.cpu().numpy(), g["train/residual1"], atol=2e-5, rtol=1e-5)
        assert torch.equal(got[1, :, 1:], h[1, :, 1:])  # pads of the length-1 sequence: the raw input, bit for bit
        for k in g.files:
            if k.startswith("after_train/residual1."):
                np.testing.assert_allclose(blk.state_dict()[k[len("after_train/residual1."):]].cpu().numpy(), g[k], atol=1e-6,
                                           rtol=1e-5, err_msg=k)
    with pytest.raises(ValueError, match=r"expected \[B, 52, L\]"):
        enc.resnet_blocks[0](h[:, :40].contiguous(), lens)


@pytest.mark.parametrize("B,Lmax,lens", [(3, 96, [96, 1, 40]), (2, 400, [400, 333]),
                                          (10, 512, [512, 1, 333, 512, 77, 500, 40, 511, 256, 129])])  # 256x192 conv tiles
def test_encoder_full_width_vs_oracle(B, Lmax, lens):
    """Real channel counts (1100 / 550, k=9, dilations 1..81) against the oracle on the same seeded inputs."""
    cfg = dict(num_labels=11, input_channels=20, output_channels=1100, kernel_size=9, dilation_base=3,
               num_resnet_blocks=5, bottleneck_factor=0.5)
    gen = torch.Generator().manual_seed(11)
    sd = random_encoder_sd(cfg, gen)
    ids = torch.randint(0, 20, (B, Lmax), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    lens_t = torch.tensor(lens)
    ref = O.proteinfer_get_embeddings({k: v.clone() for k, v in sd.items()}, x, lens_t)
    enc = make_encoder(sd, "", cfg, DEV).eval()
    for p in enc.parameters():
        p.requires_grad = False
    emb = enc.get_embeddings(x.to(DEV), lens_t.to(DEV))
    err = (emb.cpu() - ref).abs().max().item()
    assert err < 2e-4, err


# ----------------------------------------------------------------------------------------------- ProtNote eval
@pytest.mark.parametrize("fusion", ["concatenation", "concatenation_diff", "concatenation_prod", "similarity"])
def test_protnote_eval_golden(golden_dir, fusion):
    g = _g(golden_dir, f"protnote_small_{fusion}.npz")
    model, _ = make_protnote(g, DEV)
    model.eval()
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    lab = torch.from_numpy(g["label_embeddings"]).to(DEV)
    with torch.no_grad():
        model.inference_descriptions_per_label = 2
        ens, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab)
        model.inference_descriptions_per_label = 1
        raw, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab)
    np.testing.assert_allclose(raw.cpu().numpy(), g["eval/logits_raw"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(ens.cpu().numpy(), g["eval/logits_ens2"], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize("B,NL,chunk", [(8, 300, None), (5, 77, 10), (130, 33, 7)])
def test_pairhead_eval_real_width_vs_oracle(B, NL, chunk):
    """d=1024, h=3072, 3 hidden layers (base_config.yaml) on a small pair grid, vs the oracle's naive
    materialised-joint-tensor formulation."""
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(5)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    model.pair_label_chunk = chunk
    with torch.no_grad():
        out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    err = (out.cpu() - ref).abs().max().item()
    assert ref.abs().max().item() > 0.5  # logits are O(1): the tolerance is not vacuous
    assert err < 1e-3, err
    assert err < 3e-4, err


def test_zero_shot_bucketed_padding_and_label_swap():
    """BASELINE configs[4] behaviour on a small model: variable-length sequences padded to bucket sizes give
    the same logits as padding to the batch maximum, and swapping the label table (GO -> EC sized, 2
    descriptions per label, ensembled) between calls needs no model re-creation."""
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer

    gen = torch.Generator().manual_seed(9)
    ecfg = dict(num_labels=8, input_channels=20, output_channels=64, kernel_size=9, dilation_base=3,
                num_resnet_blocks=5, bottleneck_factor=0.5)
    sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
    sd.update(random_head_sd(gen, 64, 32, 16, 48, 4, 48, 3))
    enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
    model = ProtNote(protein_embedding_dim=64, label_embedding_dim=32, latent_dim=16, sequence_encoder=enc,
                     output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, inference_descriptions_per_label=2)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    lens = torch.tensor([700, 33, 1500, 128, 2048, 5])
    ids = torch.randint(0, 20, (len(lens), 2048), generator=gen)
    go = torch.randn(2 * 37, 32, generator=gen)
    ec = torch.randn(2 * 11, 32, generator=gen)

    def run(rows, lmax, labels):
        x = torch.nn.functional.one_hot(ids[rows, :lmax], 20).permute(0, 2, 1).float().contiguous()
        with torch.no_grad():
            out, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens[rows].to(DEV),
                           label_embeddings=labels.to(DEV))
        return out.cpu()

    full = run(torch.arange(6), 2048, go)
    osd = {k: v.clone() for k, v in sd.items()}
    ref = O.protnote_forward(osd, torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float(), lens, go,
                             descriptions_per_label=2)
    assert (full - ref).abs().max().item() < 5e-4
    buckets = {128: [1, 3, 5], 1024: [0], 2048: [2, 4]}
    for lmax, rows in buckets.items():
        part = run(torch.tensor(rows), lmax, go)
        assert (part - full[rows]).abs().max().item() < 2e-5, lmax
    ec_out = run(torch.arange(6), 2048, ec)  # runtime label-table swap
    ref_ec = O.protnote_forward(osd, torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float(), lens, ec,
                                descriptions_per_label=2)
    assert ec_out.shape == (6, 11) and (ec_out - ref_ec).abs().max().item() < 5e-4


@pytest.mark.parametrize("math_mode", ["f32", "bf16x3"])
def test_zero_shot_full_width_vs_oracle(math_mode):
    """BASELINE configs[4] at the REAL model width (base_config.yaml: C=1100 / 5 blocks / dil 3^i, d=1024, h=3072,
    4-layer projections, 3-layer output MLP): sequences of length 2048 / 1500 / 129 / 5 padded to 2048 and to their
    length buckets, a GO-sized-shaped table of 64 labels x 2 descriptions ensembled (ProtNote.py:308-322), then an
    EC-shaped table swapped in on the same model object (bin/test_models.py:14-23) - against the CPU oracle's naive
    formulation.  Tolerance: 5e-4 absolute on O(1) logits (north-star bound 1e-3), in both arithmetic modes."""
    import protnote_amd
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer

    gen = torch.Generator().manual_seed(21)
    ecfg = dict(num_labels=8, input_channels=20, output_channels=1100, kernel_size=9, dilation_base=3,
                num_resnet_blocks=5, bottleneck_factor=0.5)
    sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
    sd.update(random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3))
    enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
    model = ProtNote(protein_embedding_dim=1100, label_embedding_dim=1024, latent_dim=1024, sequence_encoder=enc,
                     output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, inference_descriptions_per_label=2)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    lens = torch.tensor([2048, 1500, 129, 5])
    ids = torch.randint(0, 20, (len(lens), 2048), generator=gen)
    go = torch.randn(2 * 64, 1024, generator=gen)
    ec = torch.randn(2 * 21, 1024, generator=gen)
    onehots = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float()
    for b, n in enumerate(lens):
        onehots[b, :, n:] = 0
    # the random head gives logits of mean -19 / std 8 (everything saturates in the ensembling's logit(eps=1e-7)):
    # rescale the output neuron so that the raw logits are ~N(0, 1.5^2) and the tolerance below means something
    raw = O.protnote_forward({k: v.clone() for k, v in sd.items()}, onehots, lens, go, descriptions_per_label=1)
    alpha = 1.5 / raw.std().item()
    sd["output_layer.11.bias"] = alpha * (sd["output_layer.11.bias"] - raw.mean())
    sd["output_layer.11.weight"] = alpha * sd["output_layer.11.weight"]
    model.load_state_dict(sd)
    osd = {k: v.clone() for k, v in sd.items()}
    ref_go = O.protnote_forward(osd, onehots, lens, go, descriptions_per_label=2)
    ref_ec = O.protnote_forward(osd, onehots, lens, ec, descriptions_per_label=2)
    assert 1.0 < ref_go.abs().max().item() < 9.0 and ref_go.shape == (4, 64) and ref_ec.shape == (4, 21)

    def run(rows, lmax, labels):
        x = onehots[rows][:, :, :lmax].contiguous()
        with torch.no_grad():
            out, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens[rows].to(DEV),
                           label_embeddings=labels.to(DEV))
        return out.cpu()

    protnote_amd.set_math_mode(math_mode)
    try:
        full = run(torch.arange(4), 2048, go)
        err = (full - ref_go).abs().max().item()
        assert err < 5e-4, err
        for lmax, rows in {2048: [0, 1], 256: [2], 128: [3]}.items():  # bucketed padding: same logits
            part = run(torch.tensor(rows), lmax, go)
            assert (part - ref_go[rows]).abs().max().item() < 5e-4, lmax
            assert (part - full[rows]).abs().max().item() < (2e-5 if math_mode == "f32" else 2e-4), lmax
        ec_out = run(torch.arange(4), 2048, ec)  # runtime label-table swap, no model re-creation
        assert ec_out.shape == (4, 21) and (ec_out - ref_ec).abs().max().item() < 5e-4
    finally:
        protnote_amd.set_math_mode("f32")


@pytest.mark.parametrize("fusion", ["concatenation_diff", "concatenation_prod"])
def test_pairhead_eval_fusion_variants_real_width(fusion):
    """3d-wide first layer (P, L, P-L | P.L) at d=1024 / h=3072 against the oracle's materialised joint tensor."""
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(15)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3, in_mult=3)
    B, NL = 9, 50
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f,
                             fusion=fusion)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3, feature_fusion=fusion)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    model.pair_label_chunk = 16
    with torch.no_grad():
        out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    err = (out.cpu() - ref).abs().max().item()
    assert ref.abs().max().item() > 0.5 and err < 5e-4, err


def test_map_parity_full_width():
    """mAP parity (BASELINE metric: 'mAP parity vs reference'): micro / macro average precision computed from
    the HIP logits equals the one computed from the CPU-oracle logits on the same synthetic evaluation set
    (full-width model, 96 sequences x 300 labels, targets drawn from the oracle's own probabilities so that AP
    is far from the prevalence floor)."""
    from oracle import metrics_oracle as MO
    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.utils.evaluation import DeviceAveragePrecision

    gen = torch.Generator().manual_seed(31)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 96, 300
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
    # device end to end: HIP logits -> sigmoid -> resident accumulator -> HIP AP kernels; reference: oracle logits ->
    # CPU metric oracle
    acc = DeviceAveragePrecision(NL, B, DEV)
    acc.update(torch.sigmoid(out), torch.from_numpy(y).to(DEV))
    m = acc.compute()
    mi, ma = m["map_micro"], m["map_macro"]
    pr = torch.sigmoid(ref).numpy()
    mi_ref = MO.average_precision_fast(pr.ravel(), y.ravel())
    ma_ref = MO.macro_mean([MO.average_precision_fast(pr[:, j], y[:, j]) for j in range(NL)])
    assert 0.2 < mi_ref < 0.99
    assert abs(mi - mi_ref) < 1e-4 and abs(ma - ma_ref) < 1e-4, (mi, mi_ref, ma, ma_ref)


@pytest.mark.parametrize("fusion", ["concatenation", "concatenation_prod"])
def test_save_embeddings_and_attention_pooling(golden_dir, fusion):
    """save_embeddings=True returns the reference's joint / penultimate tensors (protein-major rows) and
    LABEL_EMBEDDING_POOLING_METHOD='all' pools token embeddings with the additive attention of ProtNote.py:154-166."""
    import torch.nn.functional as F

    g = _g(golden_dir, f"protnote_small_{fusion}.npz")
    model, sd = make_protnote(g, DEV)
    model.eval()
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])
    with torch.no_grad():
        logits, emb = model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab.to(DEV),
                            save_embeddings=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["eval/logits_raw"], atol=5e-4, rtol=1e-4)
    # oracle: penultimate activations of the output MLP on the materialised joint tensor
    P_e, L_e = torch.from_numpy(g["eval/P_e"]), torch.from_numpy(g["eval/L_e"])
    joint = O.joint_embeddings(P_e, L_e, fusion)
    np.testing.assert_allclose(emb["joint_embeddings"].numpy(), joint.numpy(), atol=1e-4, rtol=1e-4)
    hsd = {k: v.clone() for k, v in sd.items()}
    hid = joint
    lin = O._linear_indices(hsd, "output_layer.")
    for i in lin[:-1]:
        hid = F.relu(O._bn(F.linear(hid, hsd[f"output_layer.{i}.weight"]), hsd, f"output_layer.{i + 1}.", False, 1e-5, 0.1))
    assert emb["output_layer_embeddings"].shape == hid.shape
    np.testing.assert_allclose(emb["output_layer_embeddings"].numpy(), hid.numpy(), atol=5e-4, rtol=1e-3)

    # additive attention pooling
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(4)
    m2 = ProtNote(protein_embedding_dim=8, label_embedding_dim=24, latent_dim=8, label_embedding_pooling_method="all",
                  output_mlp_hidden_dim_scale_factor=2, output_mlp_num_layers=2, projection_head_num_layers=2,
                  projection_head_hidden_dim_scale_factor=2).to(DEV).eval()
    hs = torch.randn(7, 11, 24, generator=gen)
    mask = (torch.rand(7, 11, generator=gen) < 0.7).long()
    mask[:, 0] = 1
    w, b = m2.raw_attn_scorer.weight.detach().cpu(), m2.raw_attn_scorer.bias.detach().cpu()
    scores = (hs @ w.T).squeeze(-1) + b
    ref = torch.bmm(torch.softmax(scores.masked_fill(mask == 0, float("-inf")), -1).unsqueeze(1), hs).squeeze(1)
    out = m2.additive_attention(hs.to(DEV), mask.to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-4)


def test_label_projection_cache_and_named_tables(golden_dir):
    """Eval-mode L_e = W_l(label table) is cached per table object and label tables are swapped by name
    (ProtNote._label_projection_eval / set_label_table; the reference recomputes W_l(L_f) on every call,
    ProtNote.py:192-196,270-271): logits bit-identical with and without the cache; a hit costs no W_l projection; an
    in-place edit of the table, an optimiser step through the flat weight buffer and a train()/eval() switch each
    invalidate it."""
    from protnote_amd.models.train_path import head_parameters
    from protnote_amd.utils.optim import FusedClipAdam

    g = _g(golden_dir, "protnote_small_concatenation.npz")
    model, sd = make_protnote(g, DEV)
    model.inference_descriptions_per_label = 2
    model.eval()
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    go = torch.from_numpy(g["label_embeddings"]).to(DEV)
    ec = (go.flip(0) * 0.5 + 0.1).contiguous()
    calls = []
    real = model._project_eval
    model._project_eval = lambda seq, t: (calls.append(seq is model.W_l), real(seq, t))[1]

    def fwd(table):
        with torch.no_grad():
            return model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=table)[0]

    a0 = fwd(go)
    a1 = fwd(go)
    assert sum(calls) == 1 and torch.equal(a0, a1)               # second call: no W_l projection
    model.label_projection_cache_size = 0                          # cache off: the reference's behaviour
    assert torch.equal(fwd(go), a0) and sum(calls) == 2
    model.label_projection_cache_size = 4
    ref = O.protnote_forward(sd, x.cpu(), lens.cpu(), go.cpu(), fusion="concatenation", descriptions_per_label=2)
    assert (a0.cpu() - ref).abs().max().item() < 5e-4
    # GO -> EC -> GO by name: one projection per table, then lookups only
    model.set_label_table("GO", go)
    model.set_label_table("EC", ec)
    n0 = sum(calls)
    b_go, b_ec, b_go2, b_ec2 = fwd("GO"), fwd("EC"), fwd("GO"), fwd("EC")
    assert sum(calls) == n0 + 2 and torch.equal(b_go, a0) and torch.equal(b_go2, a0) and torch.equal(b_ec, b_ec2)
    assert not torch.equal(b_ec, a0)
    with pytest.raises(KeyError):
        fwd("nope")
    # an in-place edit of the table bumps its version counter -> recomputed
    n0 = sum(calls)
    ec.mul_(2.0)
    c = fwd("EC")
    assert sum(calls) == n0 + 1 and not torch.equal(c, b_ec)
    assert torch.equal(c, fwd(ec.clone()))                         # a fresh tensor object: plain recompute, same logits
    # an optimiser step rewrites W_l through the flat buffer (no tensor version changes): generation counter
    for n_, q_ in model.named_parameters():
        if n_.startswith("sequence_encoder"):
            q_.requires_grad = False
    opt = FusedClipAdam(head_parameters(model), lr=1e-2, max_norm=None)
    before = fwd("GO")
    opt.flat_g.fill_(1e-3)
    opt.step()
    n0 = sum(calls)
    after = fwd("GO")
    assert sum(calls) == n0 + 1 and not torch.equal(after, before)
    # train()/eval() switches clear the cache (train-mode forwards move W_l's BatchNorm buffers in place)
    model.train()
    model.eval()
    n0 = sum(calls)
    assert torch.equal(fwd("GO"), after) and sum(calls) == n0 + 1


def test_full_size_eval_properties():
    """BASELINE configs[1] size (B=256, L=512, N_L=32102, full-width model) through size-independent properties
    (the oracle cannot run this size): in eval mode a pair's logit depends only on that protein and that label, so
    (i) a re-ordered sub-batch against a label subset reproduces the corresponding entries of the full logits,
    (ii) duplicating every label row and ensembling (logit(mean(sigmoid))) returns the same logits,
    (iii) all 8.2 M logits are finite and not constant."""
    from bench import build_model, synthetic_batch

    model = build_model(torch.device(DEV), unit_scale_weights=True).eval()
    B, L, NL = 256, 512, 32102
    batch = synthetic_batch(B, L, NL, torch.device(DEV), seed=123)
    # ragged lengths inside the fixed L=512 padding
    gen = torch.Generator().manual_seed(1)
    lens = torch.randint(64, L + 1, (B,), generator=gen).to(DEV)
    x = batch["sequence_onehots"]
    lab = batch["label_embeddings"]
    with torch.no_grad():
        full, _ = model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab)
        assert full.shape == (B, NL) and bool(torch.isfinite(full).all()) and float(full.std()) > 0.1
        rows = torch.tensor([255, 3, 128, 17, 64, 200, 1, 99], device=DEV)
        cols = torch.arange(31000, 31500, device=DEV)
        sub, _ = model(sequence_onehots=x[rows].contiguous(), sequence_lengths=lens[rows],
                       label_embeddings=lab[cols].contiguous())
        assert (sub - full[rows][:, cols]).abs().max().item() < 2e-5
        # ... and against the REFERENCE ALGORITHM: eval logits are pair-local (ProtNote.py:270-309), so the CPU oracle on
        # these 8 proteins x 500 labels (naive joint tensor, explicit padding masks) pins the same entries of the
        # full-size grid - the 256-protein batch, 63 label chunks and every tile position they sit in
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        ref = O.protnote_forward(sd, x[rows].cpu(), lens[rows].cpu(), lab[cols].cpu(), fusion="concatenation")
        assert float(ref.std()) > 0.1
        assert (full[rows][:, cols].cpu() - ref).abs().max().item() < 5e-4
        assert (sub.cpu() - ref).abs().max().item() < 5e-4
        model.inference_descriptions_per_label = 2
        dup = lab[cols].repeat_interleave(2, dim=0).contiguous()
        ens, _ = model(sequence_onehots=x[rows].contiguous(), sequence_lengths=lens[rows], label_embeddings=dup)
        inside = sub.abs() < 12  # torch.special.logit(eps=1e-7) clamps |logit| at 16.1 (reference :313-322)
        assert int(inside.sum()) > 100 and (ens - sub)[inside].abs().max().item() < 2e-3
        assert (ens - sub.clamp(-16.1181, 16.1181)).abs().max().item() < 0.5
        model.inference_descriptions_per_label = 1


def test_encoder_long_sequence_vs_oracle(golden_dir):
    """MAX_SEQUENCE_LENGTH-sized input (L = 10000, base_config.yaml:79) next to a length-1 sequence: every dilation's
    halo, the position-index arithmetic and the masked mean at the largest supported length."""
    g = _g(golden_dir, "encoder_small.npz")
    sd = O.as_torch_sd(g, "sd/")
    enc = make_encoder(sd, "", npz_cfg(g, "cfg_"), DEV).eval()
    for p in enc.parameters():
        p.requires_grad = False
    gen = torch.Generator().manual_seed(2)
    lens = torch.tensor([10000, 1, 4097])
    ids = torch.randint(0, 20, (3, 10000), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    ref = O.proteinfer_get_embeddings(sd, x, lens)
    out = enc.get_embeddings(x.to(DEV), lens.to(DEV)).cpu()
    assert (out - ref).abs().max().item() < 2e-4


def test_device_batch_assembly_matches_cpu_collator(golden_dir):
    """collate_to_device (uint8 ids + offsets -> one-hots built in HBM) == the reference-layout CPU collator, bit-exact."""
    from protnote_amd.data.collators import collate_to_device, collate_variable_sequence_length

    gen = torch.Generator().manual_seed(8)
    lab = torch.randn(9, 16, generator=gen)
    cnt = torch.randint(1, 9, (9,), generator=gen)
    batch = []
    for i, n in enumerate([33, 1, 700, 128, 64]):
        ids = torch.randint(0, 20, (n,), generator=gen)
        batch.append({"sequence_onehots": torch.nn.functional.one_hot(ids, 20).T.float(), "sequence_id": f"S{i}",
                      "sequence_length": torch.tensor(n), "label_multihots": (torch.rand(9, generator=gen) < 0.3).long(),
                      "label_embeddings": lab, "label_token_counts": cnt})
    for kw in ({}, {"label_sample_size": 4}, {"in_batch_sampling": True}):
        ref = collate_variable_sequence_length(batch, **kw)
        dev = collate_to_device(batch, DEV, **kw)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert dev[k].dtype == v.dtype and dev[k].shape == v.shape and torch.equal(dev[k].cpu(), v), k
            else:
                assert dev[k] == v


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("M,N,K", [(65536, 256, 32), (70001, 512, 96), (66000, 3072, 3072)])
def test_gemm_nt_lds_dma_vs_f64_and_register_engine(M, N, K, affine):
    """The LDS-DMA kernel (gemm_dma.hpp; tall pair-grid shapes: M >= 65536, N % 256 == 0, K % 32 == 0) against an f64
    reference, and against the register-staged engine of gemm_engine.hpp: the same products accumulate in the same
    order, so the two must agree BIT FOR BIT (ragged last row tile included)."""
    from protnote_amd import _lib as L

    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) + torch.arange(N)[:, None] * 1e-3).to(DEV)  # asymmetric
    s = (torch.rand(K, generator=g) + 0.5).to(DEV) if affine else None
    t = torch.randn(K, generator=g).to(DEV) if affine else None
    act = torch.relu(A.double() * s.double() + t.double()) if affine else A.double()
    ref = act @ W.double().T

    def run(dma):
        L.check(L.lib().pn_set_f32_dma(dma))
        out = torch.full((M, N), float("nan"), device=DEV)
        L.check(L.lib().pn_gemm_nt(L.ptr(A), K, L.ptr(W), K, L.ptr(out), N, M, N, K, None, L.ptr(s), L.ptr(t), None, None,
                                   0, None, 0, L.stream_ptr()))
        torch.cuda.synchronize()
        return out

    try:
        got, old = run(1), run(0)
    finally:
        L.lib().pn_set_f32_dma(1)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() <= 2e-6 * scale * max(1, K ** 0.5)
    assert torch.equal(got, old)
    assert torch.equal(run(1), got)  # and run to run


def test_pairhead_eval_lds_dma_grid_vs_oracle():
    """f32, full-width head on a 100 x 660 pair grid (66 000 rows: the LDS-DMA kernels' territory, an odd batch size so a
    256-row tile spans several labels, and a ragged last row tile) against the oracle's materialised joint tensor - the
    big kernels against the reference algorithm directly, not only against the register-staged engine."""
    from protnote_amd.models.ProtNote import ProtNote

    gen = torch.Generator().manual_seed(41)
    sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
    B, NL = 100, 660
    P_f = torch.randn(B, 1100, generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, None, None, lab, sequence_embeddings=P_f)
    model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4,
                     projection_head_hidden_dim_scale_factor=3)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    model.pair_label_chunk = NL  # one chunk of 66 000 rows
    with torch.no_grad():
        out, _ = model(sequence_embeddings=P_f.to(DEV), label_embeddings=lab.to(DEV))
    err = (out.cpu() - ref).abs().max().item()
    assert ref.abs().max().item() > 0.5 and err < 5e-4, err


@pytest.mark.parametrize("train_bn", [False, True])
@pytest.mark.parametrize("B,Lmax,lens", [(10, 450, [450, 1, 333, 450, 77, 449, 40, 5, 256, 129]),
                                          (3, 2048, [2048, 1500, 129])])
def test_encoder_conv_dma_bit_identical(B, Lmax, lens, train_bn):
    """The all-LDS-DMA convolution kernel (gemm_conv_dma.hpp: activation staged with guard rows, re-laid weights,
    256x192 tiles) against the register-staged tap-gather engine: same products in the same order, padding adds exact
    zeros -> embeddings (and, in train mode, the BatchNorm running statistics) must agree BIT FOR BIT; and against the
    oracle.  Lengths include 1, < 4 * dilation and the full pad length; L = 2048 is the zero-shot bucket."""
    from protnote_amd import _lib as L

    cfg = dict(num_labels=11, input_channels=20, output_channels=1100, kernel_size=9, dilation_base=3,
               num_resnet_blocks=5, bottleneck_factor=0.5)
    gen = torch.Generator().manual_seed(5)
    sd = random_encoder_sd(cfg, gen)
    ids = torch.randint(0, 20, (B, Lmax), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    x[0, :, lens[0]:] = 7.0  # garbage in the padding must not leak
    lens_t = torch.tensor(lens)

    def run(dma):
        L.check(L.lib().pn_set_f32_dma(dma))
        enc = make_encoder({k: v.clone() for k, v in sd.items()}, "", cfg, DEV)
        enc.train(train_bn)
        for p in enc.parameters():
            p.requires_grad = False
        emb = enc.get_embeddings(x.to(DEV), lens_t.to(DEV))
        torch.cuda.synchronize()
        return emb, {k: v.clone() for k, v in enc.state_dict().items()}

    try:
        (got, sd_got), (old, sd_old) = run(1), run(0)
    finally:
        L.lib().pn_set_f32_dma(1)
    assert torch.equal(got, old)
    for k in sd_old:
        assert torch.equal(sd_got[k], sd_old[k]), k
    if not train_bn:
        ref = O.proteinfer_get_embeddings({k: v.clone() for k, v in sd.items()}, x, lens_t)
        assert (got.cpu() - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("C,B,Lmax,lens", [(1100, 10, 450, [450, 1, 333, 450, 77, 449, 40, 5, 256, 129]),
                                           (52, 6, 200, [200, 1, 37, 150, 199, 9])])
def test_conv1_onehot_gather_bit_identical_and_soft_input_falls_back(C, B, Lmax, lens):
    """conv1 on one-hot input runs as a gather-sum (k_conv1_gather: 9 adds per output instead of 180 multiply-adds, bound by
    the activation write).  It adds the terms the f32-MFMA chain adds, in the same order, and every other product is an
    exact zero: eval-mode embeddings are BIT-IDENTICAL to the general convolution (pn_set_conv1_gather(0)), garbage in the
    padding and an all-zero residue column included; with train-mode BatchNorm the statistics partials are summed in
    another order (running buffers within f32 rounding).  An input that is NOT one-hot (soft / mixed residues) is detected on
    the device and takes the general kernel: bit-identical to the switch being off, and equal to the oracle."""
    from protnote_amd import _lib as L

    cfg = dict(num_labels=11, input_channels=20, output_channels=C, kernel_size=9, dilation_base=3,
               num_resnet_blocks=2, bottleneck_factor=0.5)
    gen = torch.Generator().manual_seed(15)
    sd = random_encoder_sd(cfg, gen)
    ids = torch.randint(0, 20, (B, Lmax), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    x[0, :, lens[0]:] = 7.0   # garbage in the padding must not leak (and must not count as "not one-hot")
    x[1 % B, :, 0] = 0.0      # an all-zero column (unknown residue): contributes nothing
    soft = x.clone()
    soft[3 % B, :, 2] = torch.softmax(torch.randn(20, generator=gen), 0)  # one soft residue in a live position
    lens_t = torch.tensor(lens)

    def run(inp, gather, train_bn):
        L.check(L.lib().pn_set_conv1_gather(gather))
        enc = make_encoder({k: v.clone() for k, v in sd.items()}, "", cfg, DEV)
        enc.train(train_bn)
        for p in enc.parameters():
            p.requires_grad = False
        emb = enc.get_embeddings(inp.to(DEV), lens_t.to(DEV))
        torch.cuda.synchronize()
        return emb, {k: v.clone() for k, v in enc.state_dict().items()}

    try:
        g_eval, _ = run(x, 1, False)
        c_eval, _ = run(x, 0, False)
        g_soft, _ = run(soft, 1, False)
        c_soft, _ = run(soft, 0, False)
        g_tr, sd_g = run(x, 1, True)
        c_tr, sd_c = run(x, 0, True)
    finally:
        L.lib().pn_set_conv1_gather(1)
    assert torch.equal(g_eval, c_eval)
    assert torch.equal(g_soft, c_soft) and not torch.equal(g_soft, g_eval)
    ref = O.proteinfer_get_embeddings({k: v.clone() for k, v in sd.items()}, x, lens_t)
    ref_soft = O.proteinfer_get_embeddings({k: v.clone() for k, v in sd.items()}, soft, lens_t)
    assert (g_eval.cpu() - ref).abs().max().item() < 2e-4 and (g_soft.cpu() - ref_soft).abs().max().item() < 2e-4
    np.testing.assert_allclose(g_tr.cpu().numpy(), c_tr.cpu().numpy(), atol=2e-6, rtol=2e-6)
    for k in sd_c:
        np.testing.assert_allclose(sd_g[k].cpu().numpy(), sd_c[k].cpu().numpy(), atol=1e-6, rtol=1e-5, err_msg=k)
