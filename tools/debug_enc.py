import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import protnote_oracle as O
from tests.helpers import random_encoder_sd, random_head_sd
from protnote_amd.models.ProtNote import ProtNote
from protnote_amd.models.protein_encoders import ProteInfer
from protnote_amd.utils.losses import BCEWithLogitsLoss
C = int(os.environ.get("C", 1100)); NB = int(os.environ.get("NB", 5)); LM = int(os.environ.get("LM", 300))
gen = torch.Generator().manual_seed(41)
ecfg = dict(num_labels=8, input_channels=20, output_channels=C, kernel_size=9, dilation_base=3, num_resnet_blocks=NB, bottleneck_factor=0.5)
sd = {"sequence_encoder." + k: v for k, v in random_encoder_sd(ecfg, gen).items()}
sd.update(random_head_sd(gen, C, 1024, 64, 128, 2, 128, 2))
lens = torch.tensor([int(v) for v in os.environ["LENS"].split(",")]) if "LENS" in os.environ else torch.tensor([LM, 37, 1, min(222, LM), LM, min(150, LM)])
LM = int(lens.max())
B, Lmax, NL = len(lens), LM, 24
ids = torch.randint(0, 20, (B, Lmax), generator=gen)
x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous()
lab = torch.randn(NL, 1024, generator=gen); y = (torch.rand(B, NL, generator=gen) < 0.3).float()
osd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
_, l64, g64, _ = O.train_step(osd, x.double(), lens, lab.double(), y.double(), loss="BCE", apply_update=False, train_sequence_encoder=True)
enc = ProteInfer(activation=torch.nn.ReLU, **ecfg)
model = ProtNote(protein_embedding_dim=C, sequence_encoder=enc, latent_dim=64, output_mlp_hidden_dim_scale_factor=2, output_mlp_num_layers=2, projection_head_num_layers=2, projection_head_hidden_dim_scale_factor=2, train_sequence_encoder=True)
model.load_state_dict(sd); model = model.cuda().train()
logits, _ = model(sequence_onehots=x.cuda(), sequence_lengths=lens.cuda(), label_embeddings=lab.cuda())
loss = BCEWithLogitsLoss()(logits, y.cuda()); loss.backward()
print("loss", loss.item(), float(l64))
for name, p in model.named_parameters():
    if name.startswith("sequence_encoder.") and "output_layer" not in name:
        ref = g64[name]; rel = (p.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-30)
        print(f"{name:70s} rel {rel:.2e}")
