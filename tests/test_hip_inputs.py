"""Input-side forms of the ABI that carry the algorithmic bytes instead of the reference collator's dtypes (VERDICT r05
item 6): multihot targets as uint8 (pn_loss_fwd_bwd_t / pn_tp_fn_fp_t, PN_LABEL_U8), sequences as residue ids
(pn_encoder_fwd_ids, ResidueIds), label noise drawn inside the kernel (pn_label_noise_seeded / pn_uniform).  Each is held
bit for bit to the form it replaces (or, for the RNG, to its own test hook and to the distribution); the reference dtypes
(int64 multihots, f32 one-hots: collators.py:123-137) stay the defaults."""
import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O
from tests.helpers import make_encoder, make_protnote, npz_cfg, random_encoder_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ------------------------------------------------------------------------------------------------ uint8 targets
@pytest.mark.parametrize("loss_name", ["BCE", "FocalLoss", "BatchWeightedBCE", "WeightedBCE"])
def test_uint8_targets_equal_int64_targets_bit_for_bit(loss_name):
    """Loss value, dlogits and the fused TP / FN / FP counts from uint8 (and bool) multihots == from int64 multihots == from
    float32 multihots, on a ragged shape with saturating logits; calculate_tp_fn_fp likewise."""
    from protnote_amd.models.ProtNoteTrainer import calculate_tp_fn_fp
    from protnote_amd.utils.losses import get_loss

    g = torch.Generator().manual_seed(3)
    B, N = 37, 1003
    x = (torch.randn(B, N, generator=g) * 6).to(DEV)
    y64 = (torch.rand(B, N, generator=g) < 0.2).to(torch.int64).to(DEV)
    cfg = {"params": {"LOSS_FN": loss_name, "FOCAL_LOSS_GAMMA": 2, "FOCAL_LOSS_ALPHA": 0.25, "LABEL_SMOOTHING": 0.1}}
    lw = (torch.rand(N, generator=g) + 0.5).to(DEV)
    res = {}
    for name, y in (("i64", y64), ("u8", y64.to(torch.uint8)), ("bool", y64.bool()), ("f32", y64.float())):
        loss_fn = get_loss(cfg, label_weights=lw, bce_pos_weight=torch.tensor(1.5))
        counts = torch.zeros(3, N, device=DEV)
        loss_fn.metric_counts = counts
        xi = x.clone().requires_grad_(True)
        loss = loss_fn(xi, y)
        loss.backward()
        res[name] = (loss.detach().clone(), xi.grad.clone(), counts.clone(), calculate_tp_fn_fp(torch.sigmoid(x), y, 0.4))
    for name in ("u8", "bool", "f32"):
        assert torch.equal(res[name][0], res["i64"][0]) and torch.equal(res[name][1], res["i64"][1]), name
        assert torch.equal(res[name][2], res["i64"][2]), name
        assert all(torch.equal(a, b) for a, b in zip(res[name][3], res["i64"][3])), name
    assert float(res["i64"][2].sum()) == B * N - float(((torch.sigmoid(x) < 0.5) & (y64 == 0)).sum())  # tp + fn + fp = all but tn


def test_typed_target_entry_points_validate_the_kind():
    from protnote_amd import _lib as L

    x = torch.zeros(2, 4, device=DEV)
    rc = L.lib().pn_tp_fn_fp_t(L.ptr(x), L.ptr(x), 7, 2, 4, 0.5, L.ptr(x), L.ptr(x), L.ptr(x), L.stream_ptr())
    assert rc != 0 and b"target_kind" in L.lib().pn_last_error()


def test_collate_to_device_uint8_multihots_and_residue_ids(golden_dir):
    """collate_to_device(multihot_dtype=torch.uint8, residue_ids=True): the same batch with 1 B multihots and a ResidueIds in
    place of the one-hots; the default call is unchanged (int64 multihots, f32 one-hots - row (a)14)."""
    from protnote_amd.data.collators import collate_to_device
    from protnote_amd.models.protein_encoders import ResidueIds

    g = torch.Generator().manual_seed(1)
    lens = [17, 40, 1, 33]
    lab, cnt = torch.randn(12, 8, generator=g), torch.randint(1, 9, (12,), generator=g)
    batch = [{"sequence_ints": torch.randint(0, 20, (n,), generator=g).numpy(), "alphabet_size": 20, "sequence_id": f"s{i}",
              "label_multihots": (torch.rand(12, generator=g) < 0.3).long(), "label_embeddings": lab,
              "label_token_counts": cnt} for i, n in enumerate(lens)]
    ref = collate_to_device(batch, DEV)
    assert ref["label_multihots"].dtype == torch.int64 and ref["sequence_onehots"].dtype == torch.float32
    got = collate_to_device(batch, DEV, multihot_dtype=torch.uint8, residue_ids=True)
    assert got["label_multihots"].dtype == torch.uint8 and torch.equal(got["label_multihots"].long(), ref["label_multihots"])
    ids = got["sequence_onehots"]
    assert isinstance(ids, ResidueIds) and ids.shape == tuple(ref["sequence_onehots"].shape)
    oh, ln = ids.to_onehots()
    assert torch.equal(oh, ref["sequence_onehots"]) and torch.equal(ln, ref["sequence_lengths"])
    assert torch.equal(got["sequence_lengths"], ref["sequence_lengths"]) and torch.equal(ids.lengths(), ref["sequence_lengths"])


# ------------------------------------------------------------------------------------------------ encoder from residue ids
def _ragged_ids(gen, lens, alphabet=20):
    from protnote_amd.models.protein_encoders import ResidueIds

    flat = torch.cat([torch.randint(0, alphabet, (n,), generator=gen) for n in lens]).to(torch.uint8)
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    return ResidueIds(flat.to(DEV), off.to(DEV), alphabet, max(lens))


@pytest.mark.parametrize("C,lens", [(52, [200, 1, 37, 150, 199, 9]), (1100, [512, 1, 333, 512, 77, 500, 40, 511, 256, 129])])
def test_encoder_from_residue_ids_bit_identical_to_onehots(C, lens):
    """pn_encoder_fwd_ids == pn_onehot_batch + pn_encoder_fwd, bit for bit: eval mode, train-mode BatchNorm (embeddings AND
    every running statistic), toy width and the real 1100 channels; and against the CPU oracle."""
    cfg = dict(num_labels=5, input_channels=20, output_channels=C, kernel_size=9, dilation_base=3, num_resnet_blocks=5,
               bottleneck_factor=0.5)
    gen = torch.Generator().manual_seed(21)
    sd = random_encoder_sd(cfg, gen)
    ids = _ragged_ids(gen, lens)
    x, ln = ids.to_onehots()
    ref = O.proteinfer_get_embeddings({k: v.clone() for k, v in sd.items()}, x.cpu(), ln.cpu())
    for training in (False, True):
        out = {}
        for route in ("onehots", "ids"):
            enc = make_encoder(sd, "", cfg, DEV)
            for p in enc.parameters():
                p.requires_grad = False
            enc.train(training)
            emb = enc.get_embeddings(x, ln) if route == "onehots" else enc.get_embeddings(ids, ids.lengths())
            out[route] = (emb, {k: v.clone() for k, v in enc.state_dict().items()})
        assert torch.equal(out["ids"][0], out["onehots"][0]), training
        for k, v in out["onehots"][1].items():
            assert torch.equal(out["ids"][1][k], v), (training, k)
        if not training:
            assert (out["ids"][0].cpu() - ref).abs().max().item() < 2e-4


def test_protnote_forward_accepts_residue_ids(golden_dir):
    """ProtNote.forward(sequence_onehots=ResidueIds): same logits as with the one-hot tensor, eval (fused kernels) and a
    train step; an id outside the alphabet is an all-zero column, as pn_onehot_batch makes it; a trainable encoder (which
    needs the one-hots for its conv1 weight gradient) takes them through ResidueIds.to_onehots."""
    from protnote_amd.models.protein_encoders import ResidueIds

    g = np.load(f"{golden_dir}/protnote_small_concatenation.npz")
    model, _ = make_protnote(g, DEV)
    for n, p in model.named_parameters():
        if n.startswith("sequence_encoder"):
            p.requires_grad = False
    gen = torch.Generator().manual_seed(4)
    ids = _ragged_ids(gen, [60, 3, 41, 59])
    ids.flat[5] = 23  # outside the 20-letter alphabet
    x, ln = ids.to_onehots()
    assert float(x[0, :, 5].sum()) == 0.0
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous().to(DEV)
    model.eval()
    with torch.no_grad():
        a, _ = model(sequence_onehots=x, sequence_lengths=ln, label_embeddings=lab)
        b, _ = model(sequence_onehots=ids, sequence_lengths=ids.lengths(), label_embeddings=lab)
    assert torch.equal(a, b)
    model.train()
    model.label_embedding_noising_alpha = 0.0
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    res = []
    for inp in (x, ids):
        model.load_state_dict(sd0)
        for p in model.parameters():
            p.grad = None
        lg, _ = model(sequence_onehots=inp, sequence_lengths=ln, label_embeddings=lab)
        lg.sum().backward()
        res.append((lg.detach().clone(), model.W_p[0].weight.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # trainable encoder: differentiable route, gradients reach conv1
    model.train_sequence_encoder = True
    for p in model.sequence_encoder.trunk_parameters():
        p.requires_grad = True
    model.load_state_dict(sd0)
    lg, _ = model(sequence_onehots=ids, sequence_lengths=ln, label_embeddings=lab)
    lg.sum().backward()
    assert model.sequence_encoder.conv1.weight.grad is not None and float(model.sequence_encoder.conv1.weight.grad.abs().max()) > 0
    assert isinstance(ids, ResidueIds)


# ------------------------------------------------------------------------------------------------ label noise in the kernel
def test_kernel_label_noise_hook_and_distribution():
    """pn_label_noise_seeded: out == L_f + (2u - 1) * scale with the u pn_uniform reports, bit for bit; u is a 24-bit uniform
    on [0, 1): mean, variance, 32-bin histogram, lag-1 correlations along both axes, and no repeats across seeds."""
    from protnote_amd import _lib as L

    rows, cols, seed, scale = 4099, 1024, 1234567, 20.0 / 32.0
    x = torch.randn(rows, cols, device=DEV)
    out = torch.empty_like(x)
    u = torch.empty_like(x)
    L.check(L.lib().pn_label_noise_seeded(L.ptr(x), seed, scale, L.ptr(out), rows, cols, L.stream_ptr()))
    L.check(L.lib().pn_uniform(seed, rows, cols, L.ptr(u), L.stream_ptr()))
    ref = torch.empty_like(x)   # the explicit-u kernel on the hook's draw: the same expression, bit for bit
    L.check(L.lib().pn_label_noise(L.ptr(x), L.ptr(u), scale, L.ptr(ref), x.numel(), L.stream_ptr()))
    assert torch.equal(out, ref)
    assert float((out.double() - (x.double() + (2.0 * u.double() - 1.0) * scale)).abs().max()) < 5e-7   # (fma vs mul + add)
    ud = u.double()
    assert float(ud.min()) >= 0.0 and float(ud.max()) < 1.0
    assert torch.equal(u * 16777216.0, torch.floor(u * 16777216.0))   # the 24-bit grid of torch's own float uniform
    n = ud.numel()
    assert abs(float(ud.mean()) - 0.5) < 5.0 * (1 / 12 / n) ** 0.5
    assert abs(float(ud.var()) - 1 / 12) < 1e-3
    hist = torch.histc(u, bins=32, min=0.0, max=1.0).double()
    chi2 = float(((hist - n / 32) ** 2 / (n / 32)).sum())
    assert chi2 < 80.0, chi2                                          # 31 degrees of freedom: P(chi2 > 80) ~ 3e-6
    c = ud - 0.5
    assert abs(float((c[:, 1:] * c[:, :-1]).mean()) * 12) < 5e-3 and abs(float((c[1:] * c[:-1]).mean()) * 12) < 5e-3
    u2 = torch.empty_like(x)
    L.check(L.lib().pn_uniform(seed + 1, rows, cols, L.ptr(u2), L.stream_ptr()))
    assert float((u2 == u).float().mean()) < 1e-4
    # the dropout streams of the same seed are other sequences
    m = torch.empty(rows, cols, device=DEV)
    L.check(L.lib().pn_dropout_mask(seed, 100, 0.5, rows, cols, L.ptr(m), L.stream_ptr()))
    assert abs(float(((u < 0.5).float() * m).mean()) - 0.25) < 5e-3


def test_train_step_with_kernel_noise_matches_oracle_given_the_same_draw(golden_dir):
    """The default RNG of the label noise is the kernel's: a train-mode forward with LABEL_EMBEDDING_NOISING_ALPHA > 0 equals
    the oracle's when the oracle is handed the uniforms pn_uniform reports for the seed the forward drew (torch.manual_seed
    governs that seed); two forwards draw different noise; label_noise_rng = "torch" takes the reference's own call."""
    from protnote_amd import _lib as L

    g = np.load(f"{golden_dir}/protnote_small_concatenation.npz")
    model, sd = make_protnote(g, DEV)
    for n, p in model.named_parameters():
        if n.startswith("sequence_encoder"):
            p.requires_grad = False
    assert model.label_noise_rng == "kernel" and model.label_embedding_noising_alpha > 0
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous()
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous()
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(99)
    with torch.no_grad():
        lg, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab.to(DEV),
                      label_token_counts=cnt.to(DEV))
    seed = model.__dict__["_pn_last_noise_seed"]
    torch.manual_seed(99)
    assert seed == int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    u = torch.empty(lab.shape, device=DEV)
    L.check(L.lib().pn_uniform(seed, lab.shape[0], lab.shape[1], L.ptr(u), L.stream_ptr()))
    ref = O.protnote_forward({k: v.clone() for k, v in sd.items()}, x, lens, lab, fusion="concatenation", training=True,
                             noise_alpha=float(model.label_embedding_noising_alpha), noise_u=u.cpu(), label_token_counts=cnt)
    assert (lg.cpu() - ref).abs().max().item() < 5e-4
    model.load_state_dict(sd0)
    with torch.no_grad():
        lg2, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab.to(DEV),
                       label_token_counts=cnt.to(DEV))
    assert model.__dict__["_pn_last_noise_seed"] != seed and not torch.equal(lg2, lg)
    model.label_noise_rng = "nonsense"
    with pytest.raises(ValueError, match="label_noise_rng"):
        model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab.to(DEV), label_token_counts=cnt.to(DEV))
