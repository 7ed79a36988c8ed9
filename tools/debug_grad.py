import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import protnote_oracle as O
from tests.helpers import random_head_sd
from protnote_amd.models.ProtNote import ProtNote
from protnote_amd.utils.losses import BCEWithLogitsLoss
B, NL = int(os.environ.get("B", 130)), int(os.environ.get("NL", 9))
gen = torch.Generator().manual_seed(21)
sd = random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3)
P_f = torch.randn(B, 1100, generator=gen); lab = torch.randn(NL, 1024, generator=gen)
y = (torch.rand(B, NL, generator=gen) < 0.2).float()
def oracle(dtype):
    ref_sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    names = O.trainable_names(ref_sd)
    leaves = {k: ref_sd[k].clone().requires_grad_(True) for k in names}
    work = dict(ref_sd); work.update(leaves)
    lg = O.protnote_forward(work, None, None, lab.to(dtype), training=True, sequence_embeddings=P_f.to(dtype))
    ls = O.bce_loss(lg, y.to(dtype))
    return dict(zip(names, torch.autograd.grad(ls, [leaves[k] for k in names])))
g64 = oracle(torch.float64); g32 = oracle(torch.float32)
model = ProtNote(output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4, projection_head_hidden_dim_scale_factor=3)
model.load_state_dict(sd); model = model.cuda().train()
logits, _ = model(sequence_embeddings=P_f.cuda(), label_embeddings=lab.cuda())
BCEWithLogitsLoss()(logits, y.cuda()).backward()
named = dict(model.named_parameters())
for k in ["output_layer.1.weight", "output_layer.5.weight", "output_layer.4.weight", "output_layer.8.weight", "W_l.9.weight", "W_p.9.weight", "W_l.12.weight", "W_p.12.weight"]:
    ref = g64[k]; gg = named[k].grad.cpu().double(); gc = g32[k].double()
    eg = (gg - ref).abs(); ec = (gc - ref).abs(); sc = ref.abs().mean()
    print(k, "frob gpu %.2e cpu %.2e" % ((gg-ref).norm()/ref.norm(), (gc-ref).norm()/ref.norm()),
          "| cols with err>1e-2*mean: gpu", int((eg > 1e-2*sc).sum()), "cpu", int((ec > 1e-2*sc).sum()),
          "| median err/mean gpu %.1e cpu %.1e" % (eg.median()/sc, ec.median()/sc), "max gpu %.1e cpu %.1e" % (eg.max()/sc, ec.max()/sc))
