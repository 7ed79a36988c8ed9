#!/usr/bin/env python3
"""Go / no-go probe for a two-product fp16 forward: activations split a = a_hi + a_lo in fp16 (22 bits),
the two h x h weights of the output MLP rounded ONCE to fp16 (11 bits).  The dominant error of that scheme is the weight
rounding, which needs no new kernel to measure: run the exact-f32 forward at BASELINE configs[1]/[2] size with the two hidden
weights rounded to fp16 and compare the 8.2 M logits with the unrounded run.  Acceptance bar of the scheme: max |dlogit| <=
5e-4 (half the north-star bound) INCLUDING the f32 noise floor (2.2e-4 vs f64), i.e. the rounding alone must stay well below."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    for unit in (True, False):
        model = build_model(dev, unit_scale_weights=unit)
        model.label_embedding_noising_alpha = 0.0
        batch = synthetic_batch(256, 512, 32102, dev, seed=9)
        names = [n for n, p in model.named_parameters() if n.startswith("output_layer.") and p.dim() == 2 and p.shape[0] == p.shape[1]]
        orig = {n: dict(model.named_parameters())[n].detach().clone() for n in names}

        def run(train):
            model.train(train)
            with torch.no_grad():
                lg, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                              label_embeddings=batch["label_embeddings"])
            return lg.double()

        res = {}
        for train in (False, True):
            for n in names:
                dict(model.named_parameters())[n].data.copy_(orig[n])
            base = run(train)
            for kind, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
                for n in names:
                    w = orig[n]
                    assert float(w.abs().max()) < 60000 and float(w.abs().max()) > 0
                    dict(model.named_parameters())[n].data.copy_(w.to(dt).float())
                got = run(train)
                d = (got - base).abs()
                res[("train" if train else "eval") + "/" + kind] = {
                    "max_abs_dlogit": d.max().item(), "rms_dlogit": d.pow(2).mean().sqrt().item(),
                    "logit_abs_max": base.abs().max().item(), "logit_std": base.std().item(),
                    "frac_above_5e-4": (d > 5e-4).double().mean().item()}
            for n in names:
                dict(model.named_parameters())[n].data.copy_(orig[n])
        out["unit_scale_weights" if unit else "default_init"] = {"rounded": names, **res}
        del model
        import protnote_amd

        protnote_amd.free_workspaces()
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
