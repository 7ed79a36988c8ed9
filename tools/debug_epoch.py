import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import protnote_oracle as O
from tests.helpers import random_encoder_sd, random_head_sd
from protnote_amd.models.ProtNote import ProtNote
from protnote_amd.models.protein_encoders import ProteInfer
from protnote_amd.models.ProtNoteTrainer import train_step
from protnote_amd.models.train_path import head_parameters
from protnote_amd.utils.losses import get_loss
from protnote_amd.utils.optim import FusedClipAdam
DEV = "cuda"
gen = torch.Generator().manual_seed(77)
ecfg = dict(num_labels=8, input_channels=20, output_channels=1100, kernel_size=9, dilation_base=3, num_resnet_blocks=5, bottleneck_factor=0.5)
sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
sd.update(random_head_sd(gen, 1100, 1024, 1024, 3072, 4, 3072, 3))
NSEQ, LMAX, NL, BS = 64, 128, 256, 16
lens_all = torch.randint(20, LMAX + 1, (NSEQ,), generator=gen)
ids = torch.randint(0, 20, (NSEQ, LMAX), generator=gen)
lab = torch.randn(NL, 1024, generator=gen); cnt = torch.randint(3, 30, (NL,), generator=gen)
y_all = (torch.rand(NSEQ, NL, generator=gen) < 0.05).to(torch.int64)
noises = [torch.rand(NL, 1024, generator=gen) for _ in range(NSEQ // BS)]
def batch(k):
    sl = slice(k * BS, (k + 1) * BS); lens = lens_all[sl]; lmax = int(lens.max())
    x = torch.nn.functional.one_hot(ids[sl, :lmax], 20).permute(0, 2, 1).float().contiguous()
    for b in range(BS): x[b, :, lens[b]:] = 0
    return x, lens, y_all[sl]
osd = {k: v.clone() for k, v in sd.items()}; st = {}
enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
model = ProtNote(sequence_encoder=enc, output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3, projection_head_num_layers=4, projection_head_hidden_dim_scale_factor=3, label_embedding_noising_alpha=20.0)
model.load_state_dict(sd); model = model.to(DEV).train()
for n, p in model.named_parameters():
    if n.startswith("sequence_encoder"): p.requires_grad = False
opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
real = torch.rand_like
keys = ["sequence_encoder.resnet_blocks.0.bn_activation_1.0.running_mean", "sequence_encoder.resnet_blocks.4.bn_activation_2.0.running_var",
        "W_p.1.running_mean", "W_p.1.running_var", "W_p.9.running_mean", "W_l.1.running_mean", "output_layer.1.running_mean", "output_layer.9.running_var", "W_p.0.weight", "W_p.12.weight"]
for k in range(3):
    x, lens, y = batch(k)
    _, l, _, _ = O.train_step(osd, x, lens, lab, y, loss="BCE", noise_alpha=20.0, noise_u=noises[k], label_token_counts=cnt, adam_state=st)
    u = noises[k].to(DEV); torch.rand_like = lambda t, *a, **kw: u.clone()
    b = {"sequence_onehots": x.to(DEV), "sequence_lengths": lens.to(DEV), "label_embeddings": lab.to(DEV), "label_token_counts": cnt.to(DEV), "label_multihots": y.to(DEV)}
    lg = float(train_step(model, loss_fn, opt, b)); torch.rand_like = real
    got = {kk: v.detach().cpu() for kk, v in model.state_dict().items()}
    print(f"step {k}: loss gpu {lg:.6f} cpu {float(l):.6f}")
    for kk in keys:
        d = (got[kk] - osd[kk]).abs()
        print(f"   {kk:70s} max|d| {d.max():.2e}  mean|d| {d.mean():.2e}  |ref| {osd[kk].abs().mean():.2e}")
