"""TRAIN_SEQUENCE_ENCODER gradient error classes at the reference width (C = 1100, 5 blocks, k = 9):
float64 oracle (ground truth, on the device) vs (a) the oracle in f32 on the CPU - the reference's own arithmetic -,
(b) the oracle in f32 on the device (stock torch / MIOpen convolutions: another summation order), (c) the HIP path.
Prints / writes per-tensor relative Frobenius errors of the encoder gradients and the embedding error.
    python tools/encoder_grad_error.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import protnote_oracle as O
from tests.helpers import random_encoder_sd, random_head_sd

DEV = "cuda"


def main():
    C = 1100
    lens = torch.tensor([300, 37, 1, 222, 300, 150])
    gen = torch.Generator().manual_seed(41)
    ecfg = dict(num_labels=8, input_channels=20, output_channels=C, kernel_size=9, dilation_base=3,
                num_resnet_blocks=5, bottleneck_factor=0.5)
    sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
    sd.update(random_head_sd(gen, C, 1024, 64, 128, 2, 128, 2))
    B, Lmax, NL = len(lens), int(lens.max()), 24
    ids = torch.randint(0, 20, (B, Lmax), generator=gen)
    x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
    lab = torch.randn(NL, 1024, generator=gen)
    y = (torch.rand(B, NL, generator=gen) < 0.3).float()

    def oracle(dtype, where):
        osd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()).to(where) for k, v in sd.items()}
        t0 = time.time()
        aux = {}
        with torch.backends.cudnn.flags(enabled=True):
            _, l, g, _ = O.train_step(osd, x.to(where, dtype), lens.to(where), lab.to(where, dtype), y.to(where, dtype),
                                      loss="BCE", apply_update=False, train_sequence_encoder=True)
        return float(l), {k: v.detach().double().cpu() for k, v in g.items()}, time.time() - t0

    l64, g64, t64 = oracle(torch.float64, DEV)
    runs = {"oracle f32, CPU (the reference's arithmetic)": oracle(torch.float32, "cpu"),
            "oracle f32, device (stock torch / MIOpen)": oracle(torch.float32, DEV)}

    from protnote_amd.models.ProtNote import ProtNote
    from protnote_amd.models.protein_encoders import ProteInfer
    from protnote_amd.utils.losses import BCEWithLogitsLoss

    from protnote_amd import _lib as L

    for tag, f64 in (("HIP path, f32-MFMA forward convolutions (pn_set_encoder_f64(0))", 0),
                     ("HIP path, f64-accumulating forward convolutions (default)", 1)):
        L.check(L.lib().pn_set_encoder_f64(f64))
        enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
        model = ProtNote(protein_embedding_dim=C, sequence_encoder=enc, latent_dim=64, output_mlp_hidden_dim_scale_factor=2,
                         output_mlp_num_layers=2, projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=2,
                         train_sequence_encoder=True)
        model.load_state_dict(sd)
        model = model.to(DEV).train()
        for rep in range(2):  # second pass timed (first one pays allocations)
            for q in model.parameters():
                q.grad = None
            torch.cuda.synchronize()
            t0 = time.time()
            logits, _ = model(sequence_onehots=x.to(DEV), sequence_lengths=lens.to(DEV), label_embeddings=lab.to(DEV))
            loss = BCEWithLogitsLoss()(logits, y.to(DEV))
            loss.backward()
            torch.cuda.synchronize()
            dt = time.time() - t0
        runs[tag] = (float(loss.detach()), {n: p.grad.double().cpu() for n, p in model.named_parameters()
                                            if p.grad is not None}, dt)
    L.check(L.lib().pn_set_encoder_f64(1))
    # cost at the bench shape: encoder training forward, B = 256 x L = 512
    gen2 = torch.Generator().manual_seed(1)
    xb = torch.nn.functional.one_hot(torch.randint(0, 20, (256, 512), generator=gen2), 20).permute(0, 2, 1).float().to(DEV)
    lb = torch.full((256,), 512, dtype=torch.int64, device=DEV)
    cost = {}
    for f64 in (0, 1):
        L.check(L.lib().pn_set_encoder_f64(f64))
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            e = model.sequence_encoder.get_embeddings(xb, lb)
            torch.cuda.synchronize()
            cost["f64" if f64 else "f32"] = (time.time() - t0) * 1e3
        del e
    L.check(L.lib().pn_set_encoder_f64(1))
    print("encoder training forward at 256 x 512 (ms):", cost)
    names = [n for n in g64 if n.startswith("sequence_encoder.") and n.endswith("weight") and "conv" in n]
    out = {"shape": {"C": C, "lens": lens.tolist(), "blocks": 5}, "loss_f64": l64, "runs": {},
           "encoder_training_forward_ms_at_256x512": cost}
    for tag, (l, g, dt) in runs.items():
        errs = {n: (g[n] - g64[n]).norm().item() / max(g64[n].norm().item(), 1e-30) for n in names}
        out["runs"][tag] = {"loss_abs_err": abs(l - l64), "seconds": dt, "worst": max(errs.values()),
                            "conv1.weight": errs["sequence_encoder.conv1.weight"], "per_tensor": errs}
        print(f"{tag:72s} loss err {abs(l - l64):.1e}  conv1.weight {errs['sequence_encoder.conv1.weight']:.2e}  "
              f"worst conv weight {max(errs.values()):.2e}")
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
