"""Shared helpers for the parity tests: build protnote_amd modules from a reference-style state dict."""
import numpy as np
import torch


def replay_label_noise(monkeypatch, draw):
    """Feed a recorded draw of the label-embedding noise (a reference run's `u`, ProtNote.py:219-240) to the twin: switch it
    from its default in-kernel counter-hash RNG to the reference's own call, torch.rand_like, and replace that call."""
    from protnote_amd.models.ProtNote import ProtNote

    monkeypatch.setattr(ProtNote, "label_noise_rng", "torch")
    monkeypatch.setattr(torch, "rand_like", draw)


def make_encoder(sd, prefix, cfg, device):
    from protnote_amd.models.protein_encoders import ProteInfer

    enc = ProteInfer(num_labels=int(cfg["num_labels"]), input_channels=int(cfg["input_channels"]),
                     output_channels=int(cfg["output_channels"]), kernel_size=int(cfg["kernel_size"]),
                     activation=torch.nn.ReLU, dilation_base=int(cfg["dilation_base"]),
                     num_resnet_blocks=int(cfg["num_resnet_blocks"]),
                     bottleneck_factor=float(cfg["bottleneck_factor"]))
    sub = {k[len(prefix):]: v.clone() for k, v in sd.items() if k.startswith(prefix)}
    enc.load_state_dict(sub)
    return enc.to(device)


def npz_cfg(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def make_protnote(g, device, fusion=None, **over):
    from protnote_amd.models.ProtNote import ProtNote
    from oracle import protnote_oracle as O

    sd = O.as_torch_sd(g, "sd/")
    ecfg = npz_cfg(g, "enc_cfg_")
    hcfg = {k: v.item() for k, v in npz_cfg(g, "head_cfg_").items()}
    enc = make_encoder(sd, "sequence_encoder.", ecfg, "cpu")
    hcfg.update(over)
    model = ProtNote(sequence_encoder=enc, label_encoder=None, feature_fusion=fusion or str(g["fusion"]), **hcfg)
    model.load_state_dict(sd)
    return model.to(device), sd


def random_encoder_sd(cfg, gen):
    """Reference-layout ProteInfer state dict with randomised BN stats so activations stay O(1)."""
    C, Cin, k = cfg["output_channels"], cfg["input_channels"], cfg["kernel_size"]
    Cb = int(np.floor(C * cfg["bottleneck_factor"]))
    sd = {}

    def conv(name, co, ci, kk):
        sd[name + ".weight"] = torch.randn(co, ci, kk, generator=gen) * (1.6 / (ci * kk) ** 0.5)
        sd[name + ".bias"] = torch.randn(co, generator=gen) * 0.2

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=gen) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=gen) * 0.3
        sd[name + ".running_mean"] = torch.randn(c, generator=gen) * 0.3
        sd[name + ".running_var"] = torch.rand(c, generator=gen) * 1.5 + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    conv("conv1", C, Cin, k)
    for i in range(cfg["num_resnet_blocks"]):
        p = f"resnet_blocks.{i}."
        bn(p + "bn_activation_1.0", C)
        conv(p + "masked_conv1", Cb, C, k)
        bn(p + "bn_activation_2.0", Cb)
        conv(p + "masked_conv2", C, Cb, 1)
    sd["output_layer.weight"] = torch.randn(cfg["num_labels"], C, generator=gen) * (1.0 / C ** 0.5)
    sd["output_layer.bias"] = torch.randn(cfg["num_labels"], generator=gen) * 0.1
    return sd


def random_head_sd(gen, pdim, ldim, d, h_proj, n_proj, h_out, n_out, in_mult=2):
    sd = {}

    def lin(name, o, i, bias=False):
        sd[name + ".weight"] = torch.randn(o, i, generator=gen) * (1.6 / i ** 0.5)
        if bias:
            sd[name + ".bias"] = torch.randn(o, generator=gen) * 0.2

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=gen) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=gen) * 0.3
        sd[name + ".running_mean"] = torch.randn(c, generator=gen) * 0.3
        sd[name + ".running_var"] = torch.rand(c, generator=gen) * 1.5 + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    for pre, din in (("W_p.", pdim), ("W_l.", ldim)):
        dims = [din] + [h_proj] * (n_proj - 1) + [d]
        for i in range(n_proj):
            lin(f"{pre}{4 * i}", dims[i + 1], dims[i])
            if i < n_proj - 1:
                bn(f"{pre}{4 * i + 1}", dims[i + 1])
    idx = 0
    for i in range(n_out):
        lin(f"output_layer.{idx}", h_out, in_mult * d if i == 0 else h_out)
        bn(f"output_layer.{idx + 1}", h_out)
        idx += 4 if i < n_out - 1 else 3
    lin(f"output_layer.{idx}", 1, h_out, bias=True)
    return sd


def config0_case():
    """BASELINE configs[0] shape at the REAL model width, fully seeded: 64 synthetic sequences (20 <= L <= 128), 256
    labels, batch 16 -> one epoch = 4 optimisation steps with label noise.  Shared by tests/golden/make_golden.py (which
    runs the REFERENCE on it and stores losses / logits / buffers / parameter samples in config0_full_width.npz) and by the
    oracle and HIP tests, so that all three see the same weights and data."""
    gen = torch.Generator().manual_seed(77)
    ecfg = dict(num_labels=8, input_channels=20, output_channels=1100, kernel_size=9, dilation_base=3,
                num_resnet_blocks=5, bottleneck_factor=0.5)
    sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
    sd.update(random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3))
    NSEQ, LMAX, NL, BS = 64, 128, 256, 16
    lens_all = torch.randint(20, LMAX + 1, (NSEQ,), generator=gen)
    ids = torch.randint(0, 20, (NSEQ, LMAX), generator=gen)
    lab = torch.randn(NL, 1024, generator=gen)
    cnt = torch.randint(3, 30, (NL,), generator=gen)
    y_all = (torch.rand(NSEQ, NL, generator=gen) < 0.05).to(torch.int64)
    noises = [torch.rand(NL, 1024, generator=gen) for _ in range(NSEQ // BS)]

    def batch(k):
        sl = slice(k * BS, (k + 1) * BS)
        lens = lens_all[sl]
        lmax = int(lens.max())  # the collator pads to the batch maximum
        x = torch.nn.functional.one_hot(ids[sl, :lmax], 20).permute(0, 2, 1).float().contiguous()
        for b in range(BS):
            x[b, :, lens[b]:] = 0
        return x, lens, y_all[sl]

    return dict(ecfg=ecfg, sd=sd, lab=lab, cnt=cnt, y_all=y_all, noises=noises, batch=batch, n_steps=NSEQ // BS, NL=NL,
                BS=BS)
