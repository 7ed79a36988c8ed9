"""Pin the CPU oracle (oracle/protnote_oracle.py) to vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerances: 2e-5 abs on O(1) activations/logits (f32 reassociation only),
exact for integer-valued counts."""
import os

import numpy as np
import pytest
import torch

from oracle import protnote_oracle as O

FUSIONS = ("concatenation", "concatenation_diff", "concatenation_prod", "similarity")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_encoder_eval_and_train(golden_dir):
    g = _load(golden_dir, "encoder_small.npz")
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    taps = {}
    emb = O.proteinfer_get_embeddings(sd, x, lens, training=False, taps=taps)
    for k in ("conv1", "block0", "block4"):
        np.testing.assert_allclose(taps[k].numpy(), g["eval/" + k], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(emb.numpy(), g["eval/embeddings"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(O.proteinfer_forward(sd, x, lens).numpy(), g["eval/logits"], atol=5e-5, rtol=1e-5)
    # train-mode BN: batch statistics + running-stat drift of the "frozen" encoder
    emb_t = O.proteinfer_get_embeddings(sd, x, lens, training=True)
    np.testing.assert_allclose(emb_t.numpy(), g["train/embeddings"], atol=5e-5, rtol=1e-5)
    after = O.as_torch_sd(g, "sd_after_train/")
    for k, v in after.items():
        np.testing.assert_allclose(sd[k].numpy(), v.numpy(), atol=1e-5, rtol=1e-5, err_msg=k)


def test_encoder_pad_invariance(golden_dir):
    g = _load(golden_dir, "encoder_small.npz")
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    x2 = torch.zeros(x.shape[0], 20, 260)
    x2[:, :, : x.shape[2]] = x
    emb = O.proteinfer_get_embeddings(sd, x2, lens)
    np.testing.assert_allclose(emb.numpy(), g["eval/embeddings_pad260"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(emb.numpy(), g["eval/embeddings"], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("fusion", FUSIONS)
def test_protnote_eval(golden_dir, fusion):
    _check_eval(_load(golden_dir, f"protnote_small_{fusion}.npz"), fusion)


def _check_eval(g, fusion):
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])
    T = float(g["head_cfg_temperature"])
    aux = {}
    raw = O.protnote_forward(sd, x, lens, lab, fusion=fusion, temperature=T, aux=aux)
    np.testing.assert_allclose(aux["P_f"].numpy(), g["eval/P_f"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(aux["P_e"].numpy(), g["eval/P_e"], atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(aux["L_e"].numpy(), g["eval/L_e"], atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(raw.numpy(), g["eval/logits_raw"], atol=1e-4, rtol=1e-5)
    ens = O.protnote_forward(sd, x, lens, lab, fusion=fusion, temperature=T, descriptions_per_label=2)
    np.testing.assert_allclose(ens.numpy(), g["eval/logits_ens2"], atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("fusion", FUSIONS)
@pytest.mark.parametrize("loss", ("BCE", "FocalLoss"))
def test_protnote_train_step(golden_dir, fusion, loss):
    _check_train(_load(golden_dir, f"protnote_small_{fusion}.npz"), fusion, loss)


def test_protnote_output_mlp_without_batchnorm(golden_dir):
    """OUTPUT_MLP_BATCHNORM: False (get_mlp ProtNote.py:337-378 with batch_norm=False: Linear layers carry biases)."""
    g = _load(golden_dir, "protnote_small_concatenation_nobn.npz")
    assert "sd/output_layer.0.bias" in g.files and "sd/output_layer.1.weight" not in g.files
    _check_eval(g, "concatenation")
    _check_train(g, "concatenation", "BCE")
    _check_train(g, "concatenation", "FocalLoss")


def _check_train(g, fusion, loss):
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous()
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous()
    y = torch.from_numpy(g["multihots"])
    u = torch.from_numpy(g["train/noise_u"])
    logits, l, grads, gn = O.train_step(
        sd, x, lens, lab, y, loss=loss, fusion=fusion, noise_alpha=float(g["head_cfg_label_embedding_noising_alpha"]),
        noise_u=u, label_token_counts=cnt, temperature=float(g["head_cfg_temperature"]))
    p = f"train_{loss}/"
    np.testing.assert_allclose(logits.numpy(), g[p + "logits"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(float(l), float(g[p + "loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(gn), float(g[p + "grad_norm"]), rtol=1e-4)
    for k in g.files:
        if k.startswith(p + "grad/"):
            name = k[len(p + "grad/"):]
            ref = g[k]
            np.testing.assert_allclose(grads[name].numpy(), ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), err_msg=name)
    for k in g.files:
        if k.startswith(p + "sd_after/"):
            name = k[len(p + "sd_after/"):]
            np.testing.assert_allclose(sd[name].numpy(), g[k], atol=2e-5, rtol=1e-4, err_msg=name)


@pytest.mark.parametrize("name,fusion", [("protnote_small_concatenation.npz", "concatenation"),
                                         ("protnote_small_concatenation_diff.npz", "concatenation_diff"),
                                         ("protnote_small_concatenation_prod.npz", "concatenation_prod"),
                                         ("protnote_small_concatenation_nobn.npz", "concatenation")])
def test_chunked_train_forward_matches_reference_golden(golden_dir, name, fusion):
    """O.train_forward_chunked (the label-chunked, multi-pass form of the train-mode forward that the GPU suite uses
    as the independent reference at BASELINE configs[2] size) reproduces the REFERENCE's own train-mode logits and
    BatchNorm running statistics on the golden inputs - ragged chunks of 3 labels."""
    g = _load(golden_dir, name)
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous()
    u = torch.from_numpy(g["train/noise_u"])
    L_f = O.noised_label_embeddings(lab, float(g["head_cfg_label_embedding_noising_alpha"]), u)
    with torch.no_grad():
        P_f = O.proteinfer_get_embeddings(sd, x, lens, True, 3, "sequence_encoder.")
        logits = O.train_forward_chunked(sd, P_f, L_f, fusion=fusion, label_chunk=3)
    np.testing.assert_allclose(logits.numpy(), g["train_BCE/logits"], atol=1e-4, rtol=1e-5)
    l = O.bce_loss(logits, torch.from_numpy(g["multihots"]).float())
    np.testing.assert_allclose(float(l), float(g["train_BCE/loss"]), rtol=1e-5)
    n = 0
    for k in g.files:
        if k.startswith("train_BCE/sd_after/") and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            key = k[len("train_BCE/sd_after/"):]
            np.testing.assert_allclose(sd[key].numpy(), g[k], atol=2e-5, rtol=1e-4, err_msg=key)
            n += 1
    assert n > 0


@pytest.mark.parametrize("name,fusion", [("protnote_small_concatenation.npz", "concatenation"),
                                         ("protnote_small_concatenation_diff.npz", "concatenation_diff"),
                                         ("protnote_small_concatenation_prod.npz", "concatenation_prod"),
                                         ("protnote_small_concatenation_nobn.npz", "concatenation")])
@pytest.mark.parametrize("loss", ["BCE", "FocalLoss"])
def test_chunked_train_gradients_match_reference_golden(golden_dir, name, fusion, loss):
    """O.train_grads_chunked (label-chunked forward + backward with the multi-pass BatchNorm backward; the independent
    reference of the full-size HIP train step on the GPU) reproduces the REFERENCE's own gradients, loss and logits on the
    golden inputs - ragged chunks of 3 labels - for every trainable head tensor."""
    g = _load(golden_dir, name)
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous()
    u = torch.from_numpy(g["train/noise_u"])
    L_f = O.noised_label_embeddings(lab, float(g["head_cfg_label_embedding_noising_alpha"]), u)
    with torch.no_grad():
        P_f = O.proteinfer_get_embeddings(sd, x, lens, True, 3, "sequence_encoder.")
    logits, l, grads = O.train_grads_chunked(sd, P_f, L_f, torch.from_numpy(g["multihots"]), fusion=fusion, label_chunk=3,
                                             loss=loss)
    p = f"train_{loss}/"
    np.testing.assert_allclose(logits.numpy(), g[p + "logits"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(float(l), float(g[p + "loss"]), rtol=1e-5)
    keys = [k for k in g.files if k.startswith(p + "grad/")]
    assert len(keys) == len(grads) and len(keys) > 20
    for k in keys:
        ref = g[k]
        np.testing.assert_allclose(grads[k[len(p + "grad/"):]].numpy(), ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), err_msg=k)


def test_oracle_config0_full_width_vs_reference(golden_dir):
    """The oracle at the REAL model width (1100-channel ProteInfer, d = 1024, h = 3072) against the reference itself:
    BASELINE configs[0] shape (64 sequences, 256 labels, batch 16, one epoch = 4 optimisation steps with label noise, ONE Adam
    across the epoch) run through the reference's object graph by tests/golden/make_golden.py on tests.helpers.config0_case().
    Loss trajectory, gradient norms, step-1 logits, every BatchNorm buffer and the trained parameters after the epoch."""
    from tests.helpers import config0_case

    g = _load(golden_dir, "config0_full_width.npz")
    c = config0_case()
    sd = {k: v.clone() for k, v in c["sd"].items()}
    # inference at the real width (eval-mode BatchNorm, two descriptions per label ensembled), before any training
    x0, lens0, _ = c["batch"](0)
    np.testing.assert_allclose(O.proteinfer_get_embeddings(sd, x0, lens0, prefix="sequence_encoder.").numpy(), g["eval0/P_f"],
                               atol=2e-5, rtol=1e-5)
    raw = O.protnote_forward(sd, x0, lens0, c["lab"], fusion="concatenation")
    np.testing.assert_allclose(raw.numpy(), g["eval0/logits_raw"], atol=2e-4, rtol=1e-5)
    ens = O.protnote_forward(sd, x0, lens0, c["lab"], fusion="concatenation", descriptions_per_label=2)
    np.testing.assert_allclose(ens.numpy(), g["eval0/logits_ens2"], atol=2e-4, rtol=1e-4)
    st = {}
    losses, norms = [], []
    for k in range(c["n_steps"]):
        x, lens, y = c["batch"](k)
        logits, l, _, gn = O.train_step(sd, x, lens, c["lab"], y, loss="BCE", noise_alpha=20.0, noise_u=c["noises"][k],
                                        label_token_counts=c["cnt"], adam_state=st)
        losses.append(float(l))
        norms.append(float(gn))
        if k == 0:
            np.testing.assert_allclose(logits.numpy(), g["step0/logits"], atol=2e-4, rtol=1e-4)
    # (tiny-batch BatchNorm + Adam amplify f32 reassociation step over step; steps 1-2 are tight)
    np.testing.assert_allclose(losses[:2], g["losses"][:2], rtol=1e-5)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-3)
    np.testing.assert_allclose(norms[:1], g["grad_norms"][:1], rtol=1e-4)
    n_buf = n_par = 0
    for key in g.files:
        if key.startswith("after/buffer/"):
            name = key[len("after/buffer/"):]
            if name.endswith("num_batches_tracked"):
                assert int(sd[name]) == int(g[key]), name
            elif name.startswith("sequence_encoder"):
                np.testing.assert_allclose(sd[name].numpy(), g[key], atol=2e-6, rtol=1e-5, err_msg=name)
            else:
                d = np.abs(sd[name].numpy() - g[key])
                assert d.mean() < 2e-3 * max(np.abs(g[key]).mean(), 1.0), name
            n_buf += 1
        elif key.startswith("after/param_head/"):
            name = key[len("after/param_head/"):]
            d = np.abs(sd[name].reshape(-1)[:256].numpy() - g[key])
            # Adam turns f32-noise-level gradients into +-lr moves of arbitrary sign: hard bound 2 lr steps, mean well below
            assert d.max() <= 2 * 3e-4 * 4 + 1e-5 and d.mean() <= 0.25 * 3e-4 * 4, (name, d.max(), d.mean())
            np.testing.assert_allclose(float(sd[name].double().norm()), float(g["after/param_norm/" + name]), rtol=2e-3)
            n_par += 1
    assert n_buf > 40 and n_par == 31


def test_losses_and_metrics(golden_dir):
    g = _load(golden_dir, "losses_metrics.npz")
    logits = torch.from_numpy(g["logits"])
    y = torch.from_numpy(g["multihots"])
    for name in ("BCE", "BCE_pw"):
        lg = logits.clone().requires_grad_(True)
        l = O.bce_loss(lg, y.float(), pos_weight=float(g[name + "/pos_weight"]))
        l.backward()
        np.testing.assert_allclose(float(l), float(g[name + "/loss"]), rtol=1e-6)
        np.testing.assert_allclose(lg.grad.numpy(), g[name + "/dlogits"], atol=1e-9, rtol=1e-5)
    for name in ("Focal", "Focal_a", "Focal_ls"):
        gamma, alpha, ls = (float(v) for v in g[name + "/params"])
        lg = logits.clone().requires_grad_(True)
        l = O.focal_loss(lg, y.float(), gamma=gamma, alpha=alpha, label_smoothing=ls)
        l.backward()
        np.testing.assert_allclose(float(l), float(g[name + "/loss"]), rtol=1e-6)
        np.testing.assert_allclose(lg.grad.numpy(), g[name + "/dlogits"], atol=1e-9, rtol=1e-5)
    for th in (0.5, 0.3):
        tp, fn, fp = O.tp_fn_fp(torch.sigmoid(logits), y, threshold=th)
        assert np.array_equal(tp.numpy(), g[f"th{th}/tp"])          # integer-valued counts: bit-exact
        assert np.array_equal(fn.numpy(), g[f"th{th}/fn"])
        assert np.array_equal(fp.numpy(), g[f"th{th}/fp"])
        np.testing.assert_array_equal(O.f1_per_label(tp, fn, fp).numpy(), g[f"th{th}/f1"])
        np.testing.assert_array_equal(O.f1_micro(tp, fn, fp).numpy(), g[f"th{th}/f1_micro"])


def test_extra_losses(golden_dir):
    """RGDBCE / BatchWeightedBCE / WeightedBCE / CBLoss (reference losses.py:58-146) - value and d/dlogits."""
    g = _load(golden_dir, "losses_extra.npz")
    logits, y = torch.from_numpy(g["logits"]), torch.from_numpy(g["multihots"]).float()
    lw, lc = torch.from_numpy(g["label_weights"]), torch.from_numpy(g["label_counts"])
    fns = {"RGDBCE": lambda x: O.rgd_bce_loss(x, y, 0.12), "RGDBCE_hot": lambda x: O.rgd_bce_loss(x, y, 5.0),
           "BatchWeightedBCE": lambda x: O.batch_weighted_bce_loss(x, y),
           "WeightedBCE": lambda x: O.weighted_bce_loss(x, y, lw), "CBLoss": lambda x: O.cb_loss(x, y, lc),
           "SupCon": lambda x: O.supcon_loss(x, y)}
    for name, fn in fns.items():
        lg = logits.clone().requires_grad_(True)
        l = fn(lg)
        l.backward()
        np.testing.assert_allclose(float(l), float(g[name + "/loss"]), rtol=1e-6, err_msg=name)
        np.testing.assert_allclose(lg.grad.numpy(), g[name + "/dlogits"], atol=1e-9, rtol=1e-5, err_msg=name)
    assert np.isnan(g["SupCon/dlogits"][3]).all() and np.isfinite(np.delete(g["SupCon/dlogits"], 3, 0)).all()


def test_protnote_train_encoder_grads(golden_dir):
    """TRAIN_SEQUENCE_ENCODER: True - the oracle's encoder gradients equal the reference's."""
    g = _load(golden_dir, "protnote_small_concatenation.npz")
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])[0::2].contiguous()
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous()
    logits, l, grads, _ = O.train_step(
        sd, x, lens, lab, torch.from_numpy(g["multihots"]), loss="BCE", noise_alpha=20.0,
        noise_u=torch.from_numpy(g["train/noise_u"]), label_token_counts=cnt, apply_update=False,
        train_sequence_encoder=True)
    np.testing.assert_allclose(float(l), float(g["train_enc_BCE/loss"]), rtol=1e-5)
    keys = [k for k in g.files if k.startswith("train_enc_BCE/grad/")]
    assert sum(k.startswith("train_enc_BCE/grad/sequence_encoder.") for k in keys) == 18
    for k in keys:
        name = k[len("train_enc_BCE/grad/"):]
        ref = g[k]
        np.testing.assert_allclose(grads[name].numpy(), ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), err_msg=name)


def test_attention_pooling_eval_and_train(golden_dir):
    """LABEL_EMBEDDING_POOLING_METHOD: all - the oracle's additive attention, the eval logits and one train step in
    which raw_attn_scorer gets a gradient, against the reference-generated vectors (noise scale alpha / sqrt(T))."""
    g = _load(golden_dir, "protnote_small_attention.npz")
    sd = O.as_torch_sd(g, "sd/")
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    hidden, mask = torch.from_numpy(g["hidden"]), torch.from_numpy(g["attention_mask"])
    np.testing.assert_allclose(O.additive_attention(sd, hidden, mask).numpy(), g["eval/pooled"], atol=2e-6, rtol=1e-5)
    lg = O.protnote_forward(sd, x, lens, hidden, attention_mask=mask)
    np.testing.assert_allclose(lg.numpy(), g["eval/logits"], atol=1e-4, rtol=1e-5)
    logits, l, grads, gn = O.train_step(
        sd, x, lens, hidden, torch.from_numpy(g["multihots"]), loss="BCE",
        noise_alpha=float(g["head_cfg_label_embedding_noising_alpha"]), noise_u=torch.from_numpy(g["train/noise_u"]),
        label_token_counts=torch.from_numpy(g["token_counts"]), attention_mask=mask)
    np.testing.assert_allclose(logits.numpy(), g["train_BCE/logits"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(float(l), float(g["train_BCE/loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(gn), float(g["train_BCE/grad_norm"]), rtol=1e-4)
    assert "raw_attn_scorer.weight" in grads and "train_BCE/grad/raw_attn_scorer.weight" in g.files
    for k in g.files:
        if k.startswith("train_BCE/grad/"):
            name = k[len("train_BCE/grad/"):]
            np.testing.assert_allclose(grads[name].numpy(), g[k], atol=1e-5 + 1e-4 * np.abs(g[k]).max(), err_msg=name)
    for k in g.files:
        if k.startswith("train_BCE/sd_after/"):
            name = k[len("train_BCE/sd_after/"):]
            np.testing.assert_allclose(sd[name].numpy(), g[k], atol=2e-5, rtol=1e-4, err_msg=name)


OPT_CASES = ("adam", "adam_frozen_head", "adamw", "sgd", "sgd_frozen_head")


def _adam_close(got, ref, name, lr, steps, atol=3e-5, rtol=2e-4, frac=0.99):
    """Adam's update is sign-like for gradients at f32-noise level: every element within 2 lr per step, `frac` of them
    tight (same rule as tests/test_hip_train.py::_assert_adam_close)."""
    diff = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    assert diff.max() <= 2 * lr * steps + atol, (name, diff.max())
    assert (diff <= atol + rtol * np.abs(ref)).mean() >= frac, (name, (diff <= atol + rtol * np.abs(ref)).mean())


@pytest.mark.parametrize("case", OPT_CASES)
def test_optimizer_branches_vs_reference_trainer(golden_dir, case):
    """_set_optimizer's branches (ProtNoteTrainer.py:199-245), pinned to an epoch the reference's own ProtNoteTrainer ran:
    which parameters stay trainable under TRAIN_PROJECTION_HEAD: False (output_layer.* frozen, W_p / W_l not - the
    reference's startswith quirk), and Adam / AdamW / SGD with WEIGHT_DECAY over 10 batches."""
    import json

    g = _load(golden_dir, "optimizer_branches.npz")
    sd = O.as_torch_sd(g, "sd/")
    params = json.loads(str(g[case + "/params_json"]))
    head = bool(params["TRAIN_PROJECTION_HEAD"])
    assert O.trainable_names(sd, False, head) == [str(n) for n in g[case + "/trainable_names"]]
    lab, cnt = torch.from_numpy(g["label_embeddings"]), torch.from_numpy(g["label_token_counts"])
    lr, n = float(params["LEARNING_RATE"]), int(g["n_batches"])
    st, losses = {}, []
    for k in range(n):
        x, lens = torch.from_numpy(g[f"batch{k}/x"]), torch.from_numpy(g[f"batch{k}/lens"])
        logits, l, grads, _ = O.train_step(sd, x, lens, lab, torch.from_numpy(g[f"batch{k}/multihots"]), loss="BCE",
                                           noise_alpha=20.0, noise_u=torch.from_numpy(g[f"batch{k}/noise_u"]),
                                           label_token_counts=cnt, clip=float(params["CLIP_VALUE"]), lr=lr, adam_state=st,
                                           train_projection_head=head, optimizer=params["OPTIMIZER"],
                                           weight_decay=float(params["WEIGHT_DECAY"]))
        losses.append(float(l))
        if k == 0:
            np.testing.assert_allclose(logits.numpy(), g[case + "/first_logits"], atol=1e-4, rtol=1e-5)
            assert set(grads) == set(O.trainable_names(sd, False, head))
    sgd = params["OPTIMIZER"] == "SGD"
    np.testing.assert_allclose(losses, g[case + "/losses"], rtol=2e-5 if sgd else 2e-3)
    for k in g.files:
        if not k.startswith(case + "/sd_after/"):
            continue
        name = k[len(case + "/sd_after/"):]
        ref, got = g[k], sd[name].detach().numpy()
        if name.endswith("num_batches_tracked"):
            continue  # the functional oracle does not count batches
        if not head and name.startswith("output_layer") and not name.endswith(("running_mean", "running_var")):
            assert np.array_equal(got, g["sd/" + name]) and np.array_equal(ref, g["sd/" + name]), name  # frozen: untouched
        elif sgd or name.endswith(("running_mean", "running_var")) or name.startswith("sequence_encoder"):
            np.testing.assert_allclose(got, ref, atol=2e-5, rtol=2e-4, err_msg=name)
        else:
            _adam_close(got, ref, name, lr, n)


HOLES = ["1layer_concatenation", "1layer_concatenation_diff", "1layer_concatenation_prod", "1layer_concatenation_nobn",
         "3layer_concatenation", "3layer_concatenation_prod", "3layer_similarity"]


@pytest.mark.parametrize("case", HOLES)
def test_config_holes_one_hidden_layer_and_save_embeddings(golden_dir, case):
    """OUTPUT_MLP_NUM_LAYERS: 1 (get_mlp with ONE hidden layer, ProtNote.py:337-378) and save_embeddings=True in eval
    AND train mode (ProtNote.py:292-302,324-332), generated by the reference itself (make_golden.py --config_holes)."""
    g = _load(golden_dir, f"config_holes_{case}.npz")
    fusion = str(g["fusion"])
    nl = int(g["head_cfg_output_mlp_num_layers"])
    sd = O.as_torch_sd(g, "sd/")
    if fusion != "similarity":
        n_lin = len(O._linear_indices(sd, "output_layer."))
        assert n_lin == nl + 1  # nl hidden layers + the output neuron
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    lab = torch.from_numpy(g["label_embeddings"])
    T = float(g["head_cfg_temperature"])
    aux = {}
    raw = O.protnote_forward(sd, x, lens, lab, fusion=fusion, temperature=T, aux=aux)
    np.testing.assert_allclose(raw.numpy(), g["eval/logits_raw"], atol=1e-4, rtol=1e-5)
    ens = O.protnote_forward(sd, x, lens, lab, fusion=fusion, temperature=T, descriptions_per_label=2)
    np.testing.assert_allclose(ens.numpy(), g["eval/logits_ens2"], atol=1e-4, rtol=1e-5)
    for mode in ("eval/", "train/"):
        empty = fusion == "similarity"
        assert bool(g[mode + "output_layer_embeddings_is_empty_list"]) == empty
        assert bool(g[mode + "joint_embeddings_is_empty_list"]) == empty
    if fusion != "similarity":
        np.testing.assert_allclose(aux["joint_embeddings"].numpy(), g["eval/joint_embeddings"], atol=5e-5, rtol=1e-5)
        np.testing.assert_allclose(aux["output_layer_embeddings"].numpy(), g["eval/output_layer_embeddings"], atol=1e-4,
                                   rtol=1e-5)
    lab1 = lab[0::2].contiguous()
    cnt = torch.from_numpy(g["label_token_counts"])[0::2].contiguous()
    y = torch.from_numpy(g["multihots"])
    u = torch.from_numpy(g["train/noise_u"])
    taux = {}
    logits, l, grads, gn = O.train_step(sd, x, lens, lab1, y, loss="BCE", fusion=fusion, noise_u=u, label_token_counts=cnt,
                                        noise_alpha=float(g["head_cfg_label_embedding_noising_alpha"]), temperature=T,
                                        aux=taux)
    np.testing.assert_allclose(logits.numpy(), g["train/logits"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(float(l), float(g["train/loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(gn), float(g["train/grad_norm"]), rtol=1e-4)
    if fusion != "similarity":
        np.testing.assert_allclose(taux["joint_embeddings"].detach().numpy(), g["train/joint_embeddings"], atol=5e-5,
                                   rtol=1e-5)
        np.testing.assert_allclose(taux["output_layer_embeddings"].detach().numpy(), g["train/output_layer_embeddings"],
                                   atol=1e-4, rtol=1e-5)
    n_grads = 0
    for k in g.files:
        if k.startswith("train/grad/"):
            name, ref = k[len("train/grad/"):], g[k]
            np.testing.assert_allclose(grads[name].numpy(), ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), err_msg=name)
            n_grads += 1
    assert n_grads >= 14
    for k in g.files:
        if k.startswith("train/sd_after/"):
            name = k[len("train/sd_after/"):]
            np.testing.assert_allclose(sd[name].numpy(), g[k], atol=2e-5, rtol=1e-4, err_msg=name)


def test_encoder_pieces_standalone(golden_dir):
    """MaskedConv1D.forward / Residual.forward called stand-alone on inputs with garbage pads (reference-generated
    encoder_pieces.npz): the oracle's masked_conv1d / residual_block restate them - bn1 on the RAW input, the residual added
    back unmasked - eval and train mode (running statistics included)."""
    g = _load(golden_dir, "encoder_pieces.npz")
    sd = O.as_torch_sd(g, "sd/")
    h, lens = torch.from_numpy(g["h"]), torch.from_numpy(g["lens"])
    p2, p1 = "resnet_blocks.2.", "resnet_blocks.1."
    got = O.masked_conv1d(h, lens, sd[p2 + "masked_conv1.weight"], sd[p2 + "masked_conv1.bias"], 9)
    np.testing.assert_allclose(got.numpy(), g["conv/block2_masked_conv1"], atol=1e-5, rtol=1e-5)
    got = O.masked_conv1d(h[:, :26].contiguous(), lens, sd[p1 + "masked_conv2.weight"], sd[p1 + "masked_conv2.bias"], 1)
    np.testing.assert_allclose(got.numpy(), g["conv/block1_masked_conv2"], atol=1e-5, rtol=1e-5)
    for blk, dil in ((1, 3), (4, 81)):
        got = O.residual_block(h, lens, {k: v.clone() for k, v in sd.items()}, f"resnet_blocks.{blk}.", dil, False)
        np.testing.assert_allclose(got.numpy(), g[f"eval/residual{blk}"], atol=2e-5, rtol=1e-5)
    work = {k: v.clone() for k, v in sd.items()}
    got = O.residual_block(h, lens, work, p1, 3, True)
    np.testing.assert_allclose(got.numpy(), g["train/residual1"], atol=2e-5, rtol=1e-5)
    assert np.abs(got.numpy()[1, :, 1:] - g["h"][1, :, 1:]).max() == 0.0  # pads of the length-1 sequence: the raw input
    for k in g.files:
        if k.startswith("after_train/residual1."):
            np.testing.assert_allclose(work[p1 + k[len("after_train/residual1."):]].numpy(), g[k], atol=1e-6, rtol=1e-5, err_msg=k)
