"""ProteInfer encoder forward alone (B=256, L=512, 1100 channels, 5 blocks), eval and train-mode BatchNorm; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (conv_a / conv_b / staging passes)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from protnote_amd.models.protein_encoders import ProteInfer

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L = int(os.environ.get("B", 256)), int(os.environ.get("L", 512))
enc = ProteInfer(32102, 20, 1100, 9, torch.nn.ReLU, 3, 5, 0.5).to(dev)
for p in enc.parameters():
    p.requires_grad = False
batch = synthetic_batch(B, L, 8, dev, seed=1)
x, lens = batch["sequence_onehots"], batch["sequence_lengths"]
out = {}
flop = 60.896e6 * B * L  # MFLOP per residue
for mode in ("eval", "train"):
    enc.train(mode == "train")
    for _ in range(3):
        enc.get_embeddings(x, lens)
    torch.cuda.synchronize()
    n = 20
    t = time.time()
    for _ in range(n):
        enc.get_embeddings(x, lens)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    out[mode] = {"ms": dt * 1e3, "tflops_whole_encoder": flop / dt / 1e12, "residues_per_s": B * L / dt}
print(json.dumps(out))
