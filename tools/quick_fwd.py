"""Scratch timing of the eval forward at the BASELINE config-2 shape (B=256, L=512, N_L=32102)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protnote_amd.models.ProtNote import ProtNote
from protnote_amd.models.protein_encoders import ProteInfer

B = int(os.environ.get("B", 256)); L = int(os.environ.get("L", 512)); NL = int(os.environ.get("NL", 32102))
dev = "cuda"
torch.manual_seed(0)
enc = ProteInfer(32102, 20, 1100, 9, torch.nn.ReLU, 3, 5, 0.5)
model = ProtNote(sequence_encoder=enc, output_mlp_hidden_dim_scale_factor=3, output_mlp_num_layers=3,
                 projection_head_num_layers=4, projection_head_hidden_dim_scale_factor=3).to(dev).eval()
for p in model.parameters():
    p.requires_grad = False
ids = torch.randint(0, 20, (B, L))
x = torch.nn.functional.one_hot(ids, 20).permute(0, 2, 1).float().contiguous().to(dev)
lens = torch.full((B,), L, dtype=torch.int64, device=dev)
lab = torch.randn(NL, 1024, device=dev)

def stage(name, fn, n=2):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(f"{name}: {dt*1e3:.1f} ms", flush=True)
    return out, dt

with torch.no_grad():
    P_f, t_enc = stage("encoder", lambda: enc.get_embeddings(x, lens))
    print("  encoder TFLOP/s:", 60.896e6 * B * L / t_enc / 1e12)
    P_e, _ = stage("W_p", lambda: model._project_eval(model.W_p, P_f))
    L_e, t_wl = stage("W_l", lambda: model._project_eval(model.W_l, lab))
    print("  W_l TFLOP/s:", 50.3e6 * NL / t_wl / 1e12)
    pairs, t_ph = stage("pairhead", lambda: model._pairhead_eval(P_e, L_e), n=1)
    print("  pairhead actual TFLOP/s:", 37.75e6 * B * NL / t_ph / 1e12, " pairs/s:", B * NL / t_ph)
    out, t_all = stage("full forward", lambda: model(sequence_onehots=x, sequence_lengths=lens, label_embeddings=lab)[0], n=1)
    print("pairs/s fwd:", B * NL / t_all, "finite:", bool(torch.isfinite(out).all()))
