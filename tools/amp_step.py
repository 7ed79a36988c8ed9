"""Two train steps at the bench shape with bf16x3 forward + bf16 backward (for counter passes; tools/pmc_amp.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, synthetic_batch
from protnote_amd import _lib
from protnote_amd.models.ProtNoteTrainer import train_step
from protnote_amd.models.train_path import head_parameters
from protnote_amd.utils.losses import get_loss
from protnote_amd.utils.optim import FusedClipAdam

dev = torch.device("cuda:0")
_lib.set_math_mode(os.environ.get("PN_MATH_MODE", "bf16x3"))
_lib.set_backward_math("bf16")
model = build_model(dev).train()
opt = FusedClipAdam(list(head_parameters(model)), lr=3e-4, max_norm=1.0)
loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
batch = synthetic_batch(256, 512, 32102, dev, seed=1000)
counts = torch.zeros(3, 32102, dtype=torch.float32, device=dev)
for _ in range(int(os.environ.get("PN_STEPS", "2"))):
    loss = train_step(model, loss_fn, opt, batch, world_size=1, counts=counts)
torch.cuda.synchronize()
print("loss", float(loss))
