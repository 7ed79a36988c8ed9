"""Two eval forwards at the bench shape (B = 256, L = 512, 32 102 labels) with the AMP-class forward (bf16x3 base arithmetic,
forward_math = bf16, materialised-operand route) - the workload of tools/pmc_amp_fwd.sh's counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, synthetic_batch
from protnote_amd import _lib

dev = torch.device("cuda:0")
_lib.set_math_mode("bf16x3")
_lib.set_forward_math("bf16")
model = build_model(dev).eval()
batch = synthetic_batch(256, 512, 32102, dev, seed=1000)
with torch.no_grad():
    for _ in range(int(os.environ.get("PN_STEPS", "2"))):
        lg, _ = model(sequence_onehots=batch["sequence_onehots"], sequence_lengths=batch["sequence_lengths"],
                      label_embeddings=batch["label_embeddings"])
torch.cuda.synchronize()
print("logit std", float(lg.std()))
