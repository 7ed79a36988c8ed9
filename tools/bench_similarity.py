"""Train step / eval forward with FEATURE_FUSION: similarity (cosine logits / temperature) at the bench shape."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from protnote_amd.models.ProtNote import ProtNote
from protnote_amd.models.protein_encoders import ProteInfer
from protnote_amd.models.ProtNoteTrainer import train_step
from protnote_amd.models.train_path import head_parameters
from protnote_amd.utils.losses import get_loss
from protnote_amd.utils.optim import FusedClipAdam

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = ProteInfer(32102, 20, 1100, 9, torch.nn.ReLU, 3, 5, 0.5)
model = ProtNote(sequence_encoder=enc, projection_head_num_layers=4, projection_head_hidden_dim_scale_factor=3,
                 feature_fusion="similarity", label_embedding_noising_alpha=20.0).to(dev).train()
B, L, NL = 256, 512, 32102
batch = synthetic_batch(B, L, NL, dev, seed=1)
loss_fn = get_loss({"params": {"LOSS_FN": "FocalLoss", "FOCAL_LOSS_GAMMA": 2, "FOCAL_LOSS_ALPHA": -1, "LABEL_SMOOTHING": 0.0}},
                   bce_pos_weight=torch.tensor(1.0))
opt = FusedClipAdam(head_parameters(model), lr=3e-4, max_norm=1.0)
for _ in range(2):
    train_step(model, loss_fn, opt, batch)
torch.cuda.synchronize()
t = time.time()
n = 10
for _ in range(n):
    loss = train_step(model, loss_fn, opt, batch)
torch.cuda.synchronize()
dt = (time.time() - t) / n
model.eval()
with torch.no_grad():
    for _ in range(2):
        model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
              label_embeddings=batch["label_embeddings"])
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
              label_embeddings=batch["label_embeddings"])
    torch.cuda.synchronize()
    de = (time.time() - t) / n
print(json.dumps({"fusion": "similarity", "B": B, "L": L, "N_L": NL, "train_ms_per_step": dt * 1e3,
                  "train_pairs_per_s": B * NL / dt, "eval_ms": de * 1e3, "eval_pairs_per_s": B * NL / de,
                  "loss": float(loss)}))
